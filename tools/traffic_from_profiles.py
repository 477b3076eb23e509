#!/usr/bin/env python3
"""Rebuild profiles/traffic.json (the static HBM traffic bench.py attaches to its legs) from a round's committed PMC summaries.

    python tools/traffic_from_profiles.py r06

HBM bytes per launch = (2 x FETCH_SIZE + WRITE_SIZE) x 1024, the gfx950 correction of MI355X_MICROARCH.md; both counters come from separate
rocprofv3 --pmc passes (tools/profile_round.sh).  A marched step (kernel 2m) is a sequence of field_hmarch_k launches: its entry is the sum
over the launches of one step, dispatch counts divided by the count of the once-per-step variant.
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# profile tag -> (bench.py's kernel-variant prefix, grid edge, kernel family in the summary, marched?)
SHAPES = [
    ("cosetp_f8_fp16", "field_cosetp_k<nt2,mx2,my2,flat,noclamp> 15 columns", 256, "field_cosetp_k", False),
    ("cosetp_f8_e4m3", "field_cosetp_k<nt2,mx2,my2,flat,noclamp,fp8corr> 15 columns", 256, "field_cosetp_k", False),
    ("toep_f1", "field_toep_k<mx2,my2,flat,noclamp,fp8corr> 1 columns", 256, "field_toep_k", False),
    ("toep_c2_128", "field_toep_k<mx2,my2,flat,noclamp,fp8corr> 1 columns", 128, "field_toep_k", False),
    ("toep_c4", "field_toep_k<mx2,my2,flat,noclamp,fp8corr> 1 columns", 512, "field_toep_k", False),
    ("offaxis_f1", "field_coset_k<nt1,mx2,my2,flat,noclamp,fp8corr> 4 columns", 256, "field_coset_k", False),
    ("asym_f8", "field_cosetp_k<nt2,mx2,my2,flat,noclamp,fp8corr> 32 columns", 256, "field_cosetp_k", False),
    ("sweep64_cosetp", "field_cosetp_k<nt2,mx2,my2,flat,noclamp,fp8corr> 127 columns", 256, "field_cosetp_k", False),
    ("hmarch_f1", "field_hmarch_k<nf1,noclamp,one-sum>", 256, "field_hmarch_k", True),
    ("hmarch_f8", "field_hmarch_k<nf8,noclamp,one-sum>", 256, "field_hmarch_k", True),
]


def main():
    rnd = sys.argv[1] if len(sys.argv) > 1 else "r06"
    entries = []
    for tag, prefix, grid, family, marched in SHAPES:
        rel = f"profiles/{rnd}_{tag}_pmc_summary.json"
        with open(os.path.join(ROOT, rel)) as f:
            summ = json.load(f)
        rows = [v for k, v in summ.items() if isinstance(v, dict) and k.startswith(family) and "FETCH_SIZE" in v and "WRITE_SIZE" in v]
        if not rows:
            sys.exit(f"{rel}: no {family} row with FETCH_SIZE and WRITE_SIZE")
        per = [(2.0 * r["FETCH_SIZE"] + r["WRITE_SIZE"]) * 1024.0 for r in rows]
        if marched:
            steps = min(r["dispatches"] for r in rows)
            total = sum(b * r["dispatches"] / steps for b, r in zip(per, rows))
            src = f"{rel} (sum over the field_hmarch_k launches of one step; the row-pair spread u_texel_k, ~0.2 GB, not included)"
        else:
            if len(rows) != 1:
                sys.exit(f"{rel}: {len(rows)} {family} variants, expected one")
            total, src = per[0], rel
        entries.append({"kernel_prefix": prefix, "grid": grid, "hbm_bytes_per_launch": total, "source": src,
                        "library_built_from_commit": summ.get("library_built_from_commit")})
    out = {"note": "HBM bytes per launch = (2 FETCH_SIZE + WRITE_SIZE) x 1024 from separate rocprofv3 --pmc passes (tools/profile_%s.sh, collected by "
                   "tools/traffic_from_profiles.py); bench.py reports an entry as roofline.traffic only when the kernel variant it ran starts with "
                   "kernel_prefix on the same grid, and labels it static.  bench.py measures roofline.traffic of its headline in the run itself (child "
                   "rocprofv3 --pmc passes); this file serves the legs, --static-traffic and runs with more than one rank." % rnd,
           "entries": entries}
    with open(os.path.join(ROOT, "profiles", "traffic.json"), "w") as f:
        json.dump(out, f, indent=1)
        f.write("\n")
    for e in entries:
        print(f"{e['hbm_bytes_per_launch'] / 1e6:10.1f} MB  grid {e['grid']:3d}  {e['kernel_prefix']}")


if __name__ == "__main__":
    main()
