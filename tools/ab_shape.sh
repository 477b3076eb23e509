# GPU box: kernel 2g's two block shapes, same box, alternating: pair tables (two blocks per CU) | one super-block per stage (three blocks per CU)
export OLX_LIB_PATH=$GRAFT_REPO_ROOT/openlifu-python_amd/lib/libolx_ab.so
for rep in 1 2 3; do for v in pair single; do
  export OLX_COSETP_SHAPE=$v
  python bench.py --no-extras --cpu-seconds 0 --steps 400 --warmup 30 "$@" 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$v |', round(d['roofline']['kernel_ms_avg'],4), round(d['ms_per_step'],4), round(d['roofline']['frac'],4), d['config']['kernel'][-58:])"
done; done
