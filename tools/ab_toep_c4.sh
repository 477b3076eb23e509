# GPU box: same-box alternating A/B of several library builds on BASELINE configs[3] (kernel 2f, 1024 elements, 512^3):  tools/ab_toep_c4.sh REPS LIB [LIB ...]
reps=$1; shift
for rep in $(seq $reps); do for l in "$@"; do
  OLX_LIB_PATH=$GRAFT_REPO_ROOT/openlifu-python_amd/lib/$l python bench.py --foci-per-gpu 1 --no-extras --cpu-seconds 0 --steps 100 --elements 32x32 --pitch-mm 1.5 --grid 512 --spacing-mm 0.125 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('%-22s |' % '$l', round(d['roofline']['kernel_ms_avg'],4), d['config']['kernel'][-48:])"
done; done
