# GPU box experiment: are kernel 2m's writer launches faster per element when the U double buffer fits the 256 MB Infinity Cache?
# 16x16 elements: 2 x 134 MB (does not fit); 8x16: 2 x 67 MB; 8x8: 2 x 34 MB.  Per-launch durations of the ES = 16 (writer) instantiation.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for el in 16x16 8x16 8x8; do
  rm -rf gpurun_out/exp_mall_$el
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/exp_mall_$el -- python3 bench.py --medium skull --foci-per-gpu 1 --elements $el --no-extras --cpu-seconds 0 --steps 20 --warmup 3 > /dev/null 2>&1
  f=$(find gpurun_out/exp_mall_$el -name "*kernel_stats.csv" | head -1)
  echo "== $el"; grep hmarch "$f" | cut -c1-200
done
