# GPU box: kernel 2f with one against three row tiles per block on BASELINE configs[3] (same box, alternating; needs a developer library -- any -D build --
# for the OLX_EXP_TOEP_NM pin):  tools/ab_toep_nm.sh REPS [LIB]
reps=${1:-2}; lib=${2:-libolx_dev.so}
for rep in $(seq $reps); do for nm in 1 3; do
  OLX_EXP_TOEP_NM=$nm OLX_LIB_PATH=$GRAFT_REPO_ROOT/openlifu-python_amd/lib/$lib python bench.py --foci-per-gpu 1 --no-extras --cpu-seconds 0 --steps 100 --elements 32x32 --pitch-mm 1.5 --grid 512 --spacing-mm 0.125 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('nm=$nm |', round(d['roofline']['kernel_ms_avg'],4), round(d.get('mfma_useful') or 0,3), d['config']['kernel'][-60:])"
done; done
