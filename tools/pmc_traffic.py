#!/usr/bin/env python3
"""Aggregate two rocprofv3 PMC passes (--pmc FETCH_SIZE / --pmc WRITE_SIZE, each with --kernel-trace, csv output) of one
bench forward into HBM bytes per launch per kernel family -> profiles/<name>.json.

usage: pmc_traffic.py <dir of the FETCH_SIZE pass> <dir of the WRITE_SIZE pass> <out.json>
Units per MI355X_MICROARCH.md (HBM section): the counters are in KB; on gfx950 FETCH_SIZE reports half of a wide
coalesced stream, so it is doubled; WRITE_SIZE is exact for 16-byte-per-lane stores."""
import csv
import glob
import json
import os
import sys


def family(name):
    if name.startswith("void gemm_fast_kernel") or name.startswith("gemm_fast_kernel"):
        return "gemm_fast_kernel"
    for k in ("conv3x3_c288_kernel", "ffn_fused_kernel", "outer_pairs_kernel"):
        if k in name:
            return k
    if "gemm_wreg_kernel" in name:
        return "gemm_wreg_kernel"
    if "tied_logits_split_kernel" in name:
        return "tied_logits_split_kernel"
    if "tied_split_softmax_kernel" in name:
        return "tied_split_softmax_kernel"
    if "tied_logits_kernel" in name:
        return "tied_logits_kernel"
    if "tied_av_kernel" in name:
        return "tied_av_kernel"
    if "outer_fused_kernel" in name:
        return "outer_fused_kernel"
    if "gemm_bf16_kernel" in name:
        return "gemm_bf16_kernel<conv3x3>" if name.rstrip(">(GemmP) ").endswith(", 1") or ", 1>(GemmP)" in name else "gemm_bf16_kernel"
    if "gemm_f32_kernel" in name:
        return "gemm_f32_kernel"
    if "favor_attention_kernel" in name:
        return "favor_attention_kernel"
    if "layernorm" in name:
        return "layernorm"
    if "instnorm" in name:
        return "instnorm"
    return None


def read_counter(d, counter):
    sums, counts = {}, {}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                if r.get("Counter_Name") != counter:
                    continue
                fam = family(r["Kernel_Name"])
                if fam is None:
                    continue
                sums[fam] = sums.get(fam, 0.0) + float(r["Counter_Value"])
                counts[fam] = counts.get(fam, 0) + 1
    return sums, counts


def read_durations(d):
    tot, n = {}, {}
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                fam = family(r["Kernel_Name"])
                if fam is None:
                    continue
                tot[fam] = tot.get(fam, 0.0) + float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
                n[fam] = n.get(fam, 0) + 1
    return tot, n


def tree_hash():
    import hashlib
    h = hashlib.sha1()
    d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "rosettafold-pytorch_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".h")):
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:12]


def main():
    fd, wd, out = sys.argv[1:4]
    fs, fc = read_counter(fd, "FETCH_SIZE")
    ws, wc = read_counter(wd, "WRITE_SIZE")
    fams = {}
    for fam in sorted(set(fs) | set(ws)):
        nl = max(fc.get(fam, 0), wc.get(fam, 0), 1)
        rd = 2.0 * fs.get(fam, 0.0) * 1024 / nl
        wr = ws.get(fam, 0.0) * 1024 / nl
        fams[fam] = {"launches": nl, "FETCH_SIZE_KB_sum": fs.get(fam, 0.0), "WRITE_SIZE_KB_sum": ws.get(fam, 0.0),
                     "hbm_read_bytes_per_launch": rd, "hbm_write_bytes_per_launch": wr, "hbm_bytes_per_launch": rd + wr}
    json.dump({"tree": tree_hash(), "note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in two separate passes of `python bench.py --steps 1 "
                       "--warmup 0 --no-cpu-baseline --no-roofline --no-parity` (config 2, B=4); KB units; FETCH_SIZE doubled per "
                       "MI355X_MICROARCH.md (gfx950 reports half of a wide coalesced stream); per-launch averages over every "
                       "launch of the family in one forward (kernels run serialised under --pmc, so durations are not taken "
                       "from these passes); `tree` = sha1 of csrc/*.hip|*.h at collection time (bench.py refuses a stale file)",
               "families": fams}, open(out, "w"), indent=1)
    for k, v in fams.items():
        print(k, v["launches"], f"{v['hbm_bytes_per_launch'] / 1e6:.1f} MB/launch (read {v['hbm_read_bytes_per_launch'] / 1e6:.1f}, write {v['hbm_write_bytes_per_launch'] / 1e6:.1f})")


if __name__ == "__main__":
    main()
