# GPU box: kernel 2g's 32 x 32 x 16 MFMA form (developer library, OLX_FIELD_VARIANT=cosetp32) against the product's 16 x 16 x 32 form:
# alternating bench runs on one box, headline shard (256^3) and a grid where both forms are balanced (192^3).  profiles/r03_cosetp32_ab.txt
export OLX_LIB_PATH=$GRAFT_REPO_ROOT/openlifu-python_amd/lib/libolx_ab.so
for grid in 256 192; do for rep in 1 2; do for corr in fp16 fp8; do for v in cosetp32 auto; do
  if [ $v = auto ]; then unset OLX_FIELD_VARIANT; else export OLX_FIELD_VARIANT=$v; fi
  python bench.py --no-extras --cpu-seconds 0 --corrections $corr --steps 400 --warmup 50 --grid $grid 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('grid$grid $corr $v', round(d['roofline']['kernel_ms_avg'],4), round(d['ms_per_step'],4), d['config']['kernel'][:40])"
done; done; done; done
