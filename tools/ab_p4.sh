# GPU box: kernel 2g's row map with four column tiles (developer library, OLX_FIELD_VARIANT=cosetp4) against kernel 2e's NT = 4 shape:
# 64-focus sweep and off-axis 8-focus shard, alternating runs on one box.  profiles/r03_cosetp4_ab.txt
export OLX_LIB_PATH=$GRAFT_REPO_ROOT/openlifu-python_amd/lib/libolx_ab.so
for rep in 1 2; do for cfg in "--foci-per-gpu 64" "--offset-mm 1.3,0.7"; do for v in cosetp4 auto; do
  if [ $v = auto ]; then unset OLX_FIELD_VARIANT; else export OLX_FIELD_VARIANT=$v; fi
  python bench.py --no-extras --cpu-seconds 0 --steps 200 --warmup 30 $cfg 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$cfg $v', round(d['roofline']['kernel_ms_avg'],4), round(d['ms_per_step'],4), d['config']['kernel'][:44])"
done; done; done
