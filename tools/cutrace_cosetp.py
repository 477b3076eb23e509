#!/usr/bin/env python3
"""Developer tool: CU occupancy timeline of kernel 2g (field_cosetp_k) on the headline shard.
Build:  python openlifu-python_amd/build.py -DOLX_EXP_CUTRACE --out lib/libolx_CUTRACE.so ; on the GPU box:
  OLX_LIB_PATH=openlifu-python_amd/lib/libolx_CUTRACE.so python tools/cutrace_cosetp.py [fp8]
Per block and wave the kernel records HW_ID, XCC_ID and the shader cycle at entry, at its last store issued and acknowledged; this
script groups the blocks of the LAST launch by CU and reports how many blocks are resident over time and how long a freed slot idles."""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "openlifu-python_amd"))
import bench  # noqa: E402
import openlifu_amd as ol  # noqa: E402
from openlifu_amd import _native as nat, dist as od  # noqa: E402
from openlifu_amd.engine import grid_from_coords  # noqa: E402

fp8 = "fp8" in sys.argv[1:]
arr, setup, target, pattern = bench.synthetic_workload(256, 0.25)
sweep = np.array([f.get_position(units="m") for f in pattern.get_targets(target)])
shard = od.plan_foci_orbits(sweep, 8, centre_xy=(0.0, 0.0))[0]
eng = ol.get_engine(0); ctx = eng.ctx; eng.bind(arr)
ctx.bf_solve(sweep[shard], 1500.0)
origin, spacing, n = grid_from_coords(setup.get_coords())
ctx.field_plan(origin, spacing, n, 400e3, 1500.0, 1000.0, 1e5, flags=nat.OUT_PMAG | nat.OUT_INTENSITY | (0 if fp8 else nat.FIELD_FP16_CORRECTION))
for _ in range(20):
    ctx.field_launch()
ctx.sync()
print(ctx.field_variant())
lib = nat.load()
buf = np.zeros((16384, 8, 5), dtype=np.uint64)
lib.olx_exp_read_cutrace_cosetp.argtypes = [ctypes.c_void_p]
assert lib.olx_exp_read_cutrace_cosetp(buf.ctypes.data) == 0
used = buf[:, 0, 2] > 0
b = buf[used].astype(np.int64)
nb = b.shape[0]
hw, xcc = b[:, 0, 0], b[:, 0, 1] & 0xF
cu = (hw >> 8) & 0xF; sh = (hw >> 12) & 1; se = (hw >> 13) & 0x7
key = ((xcc * 8 + se) * 2 + sh) * 16 + cu
t_in = b[:, :, 2].min(axis=1); t_issue = b[:, :, 3].max(axis=1); t_ack = b[:, :, 4].max(axis=1)
t_in_last = b[:, :, 2].max(axis=1)
print(f"{nb} blocks on {len(np.unique(key))} CUs; per block (cycles, median / p10 / p90):")
def q(x): return f"{np.median(x):9.0f} {np.percentile(x, 10):9.0f} {np.percentile(x, 90):9.0f}"
print("  first wave in -> last wave in        ", q(t_in_last - t_in))
print("  first wave in -> last store issued   ", q(t_issue - t_in))
print("  last store issued -> acknowledged    ", q(t_ack - t_issue))
print("  per-wave issue -> ack                ", q((b[:, :, 4] - b[:, :, 3]).ravel()))
print("  spread of the waves' ack times       ", q(b[:, :, 4].max(axis=1) - b[:, :, 4].min(axis=1)))
# per CU: two slots; a block starts when a slot frees -> gap = start of k-th block - k-2-th smallest end so far
gaps, spans, busy = [], [], []
for k in np.unique(key):
    m = key == k
    s0 = np.sort(t_in[m]); e0 = np.sort(t_ack[m])
    order = np.argsort(t_in[m]); st = t_in[m][order]; en = t_ack[m][order]
    span = en.max() - st.min(); spans.append(span)
    busy.append((en - st).sum() / (2.0 * span))
    # slot model: each new block (from the third on) takes the slot freed by the earliest-ending resident block
    import heapq
    h = []
    for a, e in zip(st, en):
        if len(h) == 2:
            free = heapq.heappop(h); gaps.append(a - free)
        heapq.heappush(h, e)
gaps = np.array(gaps)
print(f"per CU: span {np.median(spans):.0f} cycles, blocks {nb / len(np.unique(key)):.1f}, slot occupancy (2 slots) {np.median(busy):.3f}")
print("  slot freed (last ack of the leaving block) -> next block's first wave in:", q(gaps), f" mean {gaps.mean():.0f}")
print(f"  sum of gaps per slot {gaps.sum() / (2 * len(np.unique(key))):.0f} cycles = {gaps.sum() / (2 * len(np.unique(key))) / np.median(spans):.3f} of the span")
if "--dump" in sys.argv[1:]:
    k = np.unique(key)[3]
    m = np.where(key == k)[0]
    order = m[np.argsort(t_in[m])]
    t0 = t_in[order[0]]
    print("one CU: block id, first wave in, last wave in, first/last store-issue over waves, first/last ack over waves, npos-ish (lifetime)")
    for bi in order:
        w = b[bi]
        print(f"  {np.where(used)[0][bi]:6d} in {w[:,2].min()-t0:8d} ..{w[:,2].max()-t0:8d}  issue {w[:,3].min()-t0:8d} ..{w[:,3].max()-t0:8d}  ack {w[:,4].min()-t0:8d} ..{w[:,4].max()-t0:8d}   simd/wave-slots {sorted(set(((w[:,0]>>4)&3).tolist()))}")
