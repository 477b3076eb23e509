#!/bin/bash
# GPU box: the round-6 profile set -- rocprofv3 kernel stats + PMC summaries of every BASELINE configuration's kernel as bench.py runs it, each
# summary stamped with the commit the library was built from (openlifu-python_amd/lib/libolx.so.stamp, written by build.py).
#   tools/profile_r06.sh [shape ...]      shapes: fp16 (opted out) fp8 (= the library default) f1 c2 c4 c5f1 c5f8 offaxis asym f64 (default: all); results under gpurun_out/prof_r06_*
cd "$GRAFT_REPO_ROOT" || exit 1
shapes=${@:-fp16 fp8 f1 c2 c4 c5f1 c5f8 offaxis asym f64}
for s in $shapes; do
  case $s in
    fp16)    bash tools/profile_round.sh r06_cosetp_f8_fp16 --no-extras --corrections fp16 ;;
    fp8)     bash tools/profile_round.sh r06_cosetp_f8_e4m3 --no-extras ;;
    f1)      bash tools/profile_round.sh r06_toep_f1 --no-extras --foci-per-gpu 1 ;;
    c2)      bash tools/profile_round.sh r06_toep_c2_128 --no-extras --foci-per-gpu 1 --grid 128 --spacing-mm 0.5 ;;
    c4)      bash tools/profile_round.sh r06_toep_c4 --no-extras --foci-per-gpu 1 --grid 512 --spacing-mm 0.125 --elements 32x32 --pitch-mm 1.5 --steps 100 --warmup 10 ;;
    c5f1)    bash tools/profile_round.sh r06_hmarch_f1 --no-extras --medium skull --foci-per-gpu 1 --steps 50 --warmup 5 ;;
    c5f8)    bash tools/profile_round.sh r06_hmarch_f8 --no-extras --medium skull --foci-per-gpu 8 --steps 30 --warmup 5 ;;
    offaxis) bash tools/profile_round.sh r06_offaxis_f1 --no-extras --foci-per-gpu 1 --offset-mm 1.3,0.7 ;;
    asym)    bash tools/profile_round.sh r06_asym_f8 --no-extras --offset-mm 1.3,0.7 ;;
    f64)     bash tools/profile_round.sh r06_sweep64_cosetp --no-extras --foci-per-gpu 64 --steps 100 --warmup 10 ;;
  esac
done
