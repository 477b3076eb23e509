# GPU box: kernel 2g fed from the precomputed geometry table (default) against the in-kernel table generation (OLX_GTABLE=0) and against
# the table with the former block order (OLX_GTABLE=order0), alternating runs on one box: headline shard (fp16 / fp8 corrections) and the
# whole 64-focus sweep (8 launch tiles share the table).  profiles/r04_gtable_ab.txt
for rep in 1 2 3; do for cfg in "--corrections fp16" "--foci-per-gpu 64 --steps 100"; do for v in 0 1 order0; do
  if [ $v = 1 ]; then unset OLX_GTABLE; else export OLX_GTABLE=$v; fi
  python bench.py --no-extras --cpu-seconds 0 --steps 400 --warmup 30 $cfg 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$cfg | OLX_GTABLE=$v |', round(d['roofline']['kernel_ms_avg'],4), round(d['ms_per_step'],4), round(d['roofline']['frac'],4), d['config']['kernel'][:50], d['config']['kernel'][-30:])"
done; done; done
