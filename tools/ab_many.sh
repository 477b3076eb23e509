# GPU box: same-box alternating A/B of several builds of the library on the headline shard:  tools/ab_many.sh REPS LIB [LIB ...] [-- bench args]
#   (names under openlifu-python_amd/lib/; every round runs each library once, in the order given)
reps=$1; shift
libs=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do libs+=("$1"); shift; done; [ "$1" = "--" ] && shift
for rep in $(seq $reps); do for l in "${libs[@]}"; do
  OLX_LIB_PATH=$GRAFT_REPO_ROOT/openlifu-python_amd/lib/$l python bench.py --no-extras --cpu-seconds 0 --steps 400 --warmup 30 "$@" 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('%-24s' % '$l', '|', round(d['roofline']['kernel_ms_avg'],4), round(d['ms_per_step'],4), round(d['roofline']['frac'],4), d['config']['kernel'][:50])"
done; done
