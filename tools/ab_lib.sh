# GPU box: same-box alternating A/B of two builds of the library on the headline shard:  tools/ab_lib.sh LIB_B [bench args]   (A = lib/libolx.so)
libb=$1; shift
for rep in 1 2 3; do for v in A B; do
  if [ $v = A ]; then unset OLX_LIB_PATH; else export OLX_LIB_PATH=$GRAFT_REPO_ROOT/openlifu-python_amd/lib/$libb; fi
  python bench.py --no-extras --cpu-seconds 0 --steps 400 --warmup 30 "$@" 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$v', '$libb' if '$v'=='B' else 'libolx.so', '|', round(d['roofline']['kernel_ms_avg'],4), round(d['ms_per_step'],4), round(d['roofline']['frac'],4), d['config']['kernel'][:50])"
done; done
