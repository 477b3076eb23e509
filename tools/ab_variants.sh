#!/bin/bash
# Developer A/B: run bench.py against every openlifu-python_amd/lib/libolx*.so experiment build (OLX_LIB_PATH).
#   tools/ab_variants.sh [bench args...]
cd "$(dirname "$0")/.."
for so in openlifu-python_amd/lib/libolx.so openlifu-python_amd/lib/libolx_*.so; do
  [ -f "$so" ] || continue
  OLX_LIB_PATH="$PWD/$so" python bench.py --cpu-seconds 0 "$@" 2>/dev/null |
    python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%-28s kernel %.4f ms  step %.4f ms  %s' % ('$(basename $so)', d['roofline']['kernel_ms_avg'], d['ms_per_step'], d['config']['kernel'][:60]))"
done
