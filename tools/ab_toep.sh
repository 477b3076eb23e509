# GPU box: same-box alternating A/B of several library builds on kernel 2f's three BASELINE shapes:  tools/ab_toep.sh REPS LIB [LIB ...] [-- corrections]
#   (names under openlifu-python_amd/lib/; every round runs each library once per shape, in the order given; prints kernel ms per launch)
reps=$1; shift
libs=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do libs+=("$1"); shift; done; [ "$1" = "--" ] && shift
corr=${1:-auto}
one() {  # label, lib, bench args...
  label=$1; lib=$2; shift 2
  OLX_LIB_PATH=$GRAFT_REPO_ROOT/openlifu-python_amd/lib/$lib python bench.py --foci-per-gpu 1 --no-extras --cpu-seconds 0 --corrections $corr "$@" 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('%-10s %-26s' % ('$label', '$lib'), '|', round(d['roofline']['kernel_ms_avg'],4), round(d.get('mfma_useful') or 0,3), d['config']['kernel'][:58])"
}
for rep in $(seq $reps); do for l in "${libs[@]}"; do
  one single256 $l --steps 500
  one c4 $l --steps 100 --elements 32x32 --pitch-mm 1.5 --grid 512 --spacing-mm 0.125
  one c2_128 $l --steps 1000 --grid 128 --spacing-mm 0.5
done; done
