# GPU box: same-box alternating runs of the headline shard under several settings of ONE environment variable:  tools/ab_env.sh REPS VAR VALUE [VALUE ...] [-- bench args]
#   (the value "-" leaves the variable unset)
reps=$1; var=$2; shift 2
vals=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do vals+=("$1"); shift; done; [ "$1" = "--" ] && shift
for rep in $(seq $reps); do for v in "${vals[@]}"; do
  if [ "$v" = "-" ]; then unset $var; else export $var=$v; fi
  python bench.py --no-extras --cpu-seconds 0 --steps 400 --warmup 30 "$@" 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('%-20s' % '$var=$v', '|', round(d['roofline']['kernel_ms_avg'],4), round(d['ms_per_step'],4), round(d['roofline']['frac'],4), d['config']['kernel'][:50])"
done; done
