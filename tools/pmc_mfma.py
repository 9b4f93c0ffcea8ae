#!/usr/bin/env python3
"""Matrix-pipe utilisation per kernel family from one rocprofv3 PMC pass of a bench forward
(--pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE, --kernel-trace, csv) -> profiles/<name>.json.

usage: pmc_mfma.py <dir of the pass> <out.json>
Units (MI355X_MICROARCH.md, cycle-constants table): SQ_VALU_MFMA_BUSY_CYCLES counts cycles a SIMD's matrix pipe is busy,
summed over every SIMD of the device (16 per v_mfma_f32_16x16x32_bf16); GRBM_GUI_ACTIVE is the sum over the 8 XCDs of the
cycles the kernel was resident, so wall cycles = GRBM_GUI_ACTIVE / 8 and
    mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (wall cycles * 256 CUs * 4 SIMDs)."""
import csv
import glob
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pmc_traffic import family, tree_hash  # noqa: E402

N_SIMD = 256 * 4


def main():
    d, out = sys.argv[1:3]
    acc = {}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                fam = family(r["Kernel_Name"])
                if fam is None:
                    continue
                a = acc.setdefault(fam, {})
                key = (r.get("Dispatch_Id"), r["Counter_Name"])
                a[key] = a.get(key, 0.0) + float(r["Counter_Value"])
    fams = {}
    for fam, a in sorted(acc.items()):
        tot = {}
        disp = set()
        for (did, cname), v in a.items():
            tot[cname] = tot.get(cname, 0.0) + v
            disp.add(did)
        wall = tot.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
        mf = tot.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
        fams[fam] = {"launches": len(disp), "SQ_VALU_MFMA_BUSY_CYCLES": mf, "GRBM_GUI_ACTIVE": tot.get("GRBM_GUI_ACTIVE", 0.0),
                     "SQ_BUSY_CYCLES": tot.get("SQ_BUSY_CYCLES", 0.0),
                     "mfma_util": mf / (wall * N_SIMD) if wall > 0 else None}
    json.dump({"tree": tree_hash(), "note": __doc__.strip().split("\n\n")[-1], "families": fams}, open(out, "w"), indent=1)
    for k, v in fams.items():
        u = v["mfma_util"]
        print(f"{k:32s} {v['launches']:5d} launches  mfma_util {u:.3f}" if u is not None else f"{k:32s} no GRBM_GUI_ACTIVE")


if __name__ == "__main__":
    main()
