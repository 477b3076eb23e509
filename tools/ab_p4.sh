mkdir -p gpurun_out/r3i



for rep in 1; do
for cfg in "--foci-per-gpu 64" "--offset-mm 1.3,0.7"; do
for v in auto lattice; do
  if [ $v = auto ]; then unset OLX_FIELD_VARIANT; else export OLX_FIELD_VARIANT=$v; fi
  python bench.py --no-extras --cpu-seconds 0 --steps 200 --warmup 30 $cfg 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$cfg $v', round(d['roofline']['kernel_ms_avg'],4), round(d['ms_per_step'],4), d['config']['kernel'][:44])"
done; done; done | tee gpurun_out/r3i/ab_p4.txt
