mkdir -p gpurun_out/r3g
for rep in 1 2; do
for corr in fp16; do
for v in auto cosetp16; do
  if [ $v = auto ]; then unset OLX_FIELD_VARIANT; else export OLX_FIELD_VARIANT=$v; fi
  python bench.py --no-extras --cpu-seconds 0 --corrections $corr --steps 400 --warmup 50 --grid 192 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('grid192 $corr $v', round(d['roofline']['kernel_ms_avg'],4), round(d['ms_per_step'],4), d['config']['kernel'][:40])"
done; done; done | tee gpurun_out/r3g/ab192.txt
