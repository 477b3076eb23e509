# GPU box: headline shard, same box, alternating, developer library: three fp16 products (the default), mixed corrections
# (OLX_MIXED_CORRECTION=1: lo_G x hi_W in e4m3), both corrections in e4m3 (opt-in).  profiles/r04_mixed_ab.txt
export OLX_LIB_PATH=$GRAFT_REPO_ROOT/openlifu-python_amd/lib/libolx_ab.so
for rep in 1 2 3; do for v in fp16 mixed fp8; do
  unset OLX_MIXED_CORRECTION; corr=fp16
  if [ $v = mixed ]; then export OLX_MIXED_CORRECTION=1; fi
  if [ $v = fp8 ]; then corr=fp8; fi
  python bench.py --no-extras --cpu-seconds 0 --steps 400 --warmup 30 --corrections $corr "$@" 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$v |', round(d['roofline']['kernel_ms_avg'],4), round(d['ms_per_step'],4), round(d['roofline']['frac'],4), d['config']['kernel'][29:48], d['config']['kernel'][-42:])"
done; done
