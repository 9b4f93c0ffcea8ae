#!/usr/bin/env python3
"""ISA audit (build container, no GPU): does any shipped kernel read an MFMA result too early?

On gfx950 the result registers of a v_mfma are written `passes` cycles-of-4 after issue, and NOTHING in hardware holds back
a following VALU / LDS / VMEM instruction that reads them: the wait states are software's job (hipcc's hazard recognizer
inserts s_nop).  Round 2's non-reproducible FAVOR+ kernel was exactly such a read: the denominator broadcast `__shfl`
(ds_bpermute_b32) sat at the head of the loop-exit block, the last MFMA of the accumulator at the tail of the loop body, and
hipcc had put no s_nop on that (branch) path -- whenever the co-resident wave kept the matrix pipe busy, the younger wave's
shuffle read the accumulator before the last feature block was added (tools/favor_rootcause/).

This script compiles every csrc/*.hip to gfx950 assembly (both 16-bit builds) and walks every kernel: from each v_mfma it
follows the fall-through path AND every branch target, counting wait states (one per instruction, N + 1 per `s_nop N`),
until a non-MFMA instruction READS a register of the MFMA's destination.  Fewer wait states than the table below -> reported.

    python tools/isa_hazard_scan.py [file.s ...]        # no arguments: build the assembly of the whole library first
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "rosettafold-pytorch_amd", "csrc")
# wait states an MFMA result needs before a VALU / LDS / VMEM / export read (passes + 2 .. 3 in LLVM's gfx940/gfx950 tables).
# hipcc itself puts `s_nop 7` + the reader behind a 16x16x32 MFMA in straight-line code, i.e. it treats it as an 8-pass
# instruction like 16x16x4 f32 and 32x32x16; 32x32x2 f32 is 16-pass.  The table is hipcc's own straight-line minimum for the
# 16x16x32 forms (8 wait states, measured over the 14,988 MFMAs of this library) and generous values for the others.
NEED = [(re.compile(r"v_mfma_f32_16x16x32_(bf16|f16)"), 8), (re.compile(r"v_mfma_f32_16x16x4_f32"), 11),
        (re.compile(r"v_mfma_f32_32x32x16"), 11), (re.compile(r"v_mfma_f32_32x32x2_f32"), 19), (re.compile(r"v_mfma"), 11)]
REG = re.compile(r"\b([va])(\d+)\b|\b([va])\[(\d+):(\d+)\]")
STORE_LIKE = re.compile(r"^(global_store|buffer_store|scratch_store|flat_store|ds_write|ds_add|global_atomic|exp)")


def regs(tok):
    out = set()
    for m in REG.finditer(tok):
        if m.group(1):
            out.add((m.group(1), int(m.group(2))))
        else:
            out.update((m.group(3), i) for i in range(int(m.group(4)), int(m.group(5)) + 1))
    return out


def parse(path):
    kernels, cur, name = {}, None, None
    for line in open(path):
        line = line.split(";")[0].rstrip()
        if not line.strip():
            continue
        m = re.match(r"^([A-Za-z_.$][\w.$]*):", line)
        if m:
            lab = m.group(1)
            if lab.startswith("_Z") or (cur is None and not lab.startswith(".")):
                name, cur = lab, []
                kernels[name] = cur
            elif cur is not None:
                cur.append(("label", lab, []))
            continue
        if cur is None or line.startswith("\t."):
            continue
        parts = line.strip().split(None, 1)
        op = parts[0]
        ops = [o.strip() for o in parts[1].split(",")] if len(parts) > 1 else []
        cur.append((op, line.strip(), ops))
        if op == "s_endpgm":
            cur = None
    return kernels


def scan_kernel(name, ins):
    labels = {text: i for i, (op, text, _) in enumerate(ins) if op == "label"}
    found = []
    for i, (op, text, ops) in enumerate(ins):
        if not op.startswith("v_mfma"):
            continue
        need = next(n for r, n in NEED if r.search(op))
        dest = regs(ops[0])
        # walk: (index, wait states so far)
        stack, seen = [(i + 1, 0)], set()
        while stack:
            j, ws = stack.pop()
            while j < len(ins) and ws < need:
                o, t, oo = ins[j]
                if (j, ws) in seen:
                    break
                seen.add((j, ws))
                if o == "label":
                    j += 1
                    continue
                if o.startswith("v_mfma"):
                    # an MFMA that takes the result whole as srcC chains for free; one that overwrites it ends the hazard window
                    if regs(oo[0]) & dest:
                        break
                    srcs = set().union(*[regs(x) for x in oo[1:3]]) if len(oo) > 2 else set()
                    if srcs & dest:
                        found.append((name, need, ws, text, t, "MFMA A/B operand"))
                        break
                    ws += 1
                    j += 1
                    continue
                srcs = set().union(*[regs(x) for x in (oo if STORE_LIKE.match(o) else oo[1:])]) if oo else set()
                if srcs & dest and not o.startswith("s_"):
                    found.append((name, need, ws, text, t, "read"))
                    break
                if oo and not STORE_LIKE.match(o) and (regs(oo[0]) & dest) and regs(oo[0]) >= dest:
                    break  # overwritten
                if o == "s_nop":
                    ws += int(oo[0]) + 1
                elif o == "s_endpgm":
                    break
                else:
                    ws += 1
                if o.startswith("s_cbranch") or o == "s_branch":
                    tgt = oo[0]
                    if tgt in labels:
                        stack.append((labels[tgt], ws))
                    if o == "s_branch":
                        break
                j += 1
    return found


def build_asm(tmp, jobs=8):
    """the same flags as csrc/Makefile, -S instead of -c, device code only"""
    from concurrent.futures import ThreadPoolExecutor
    work = []
    for f in sorted(os.listdir(CSRC)):
        if not f.endswith(".hip"):
            continue
        for tag, flags in (("bf16", []), ("f16", ["-DRF_H16_IS_F16"])):
            s = os.path.join(tmp, f"{f[:-4]}.{tag}.s")
            extra = ["-fno-honor-nans", "-fno-signed-zeros"] if f == "favor.hip" else []
            work.append((s, ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-I" + os.path.join(ROOT, "include"),
                             "-I" + CSRC, "-Wno-unused-value", "-S", "--cuda-device-only", os.path.join(CSRC, f), "-o", s] + flags + extra))
    with ThreadPoolExecutor(jobs) as ex:
        list(ex.map(lambda w: subprocess.run(w[1], check=True, stderr=subprocess.DEVNULL), work))
    return [w[0] for w in work]


def main():
    files = sys.argv[1:]
    tmp = None
    if not files:
        tmp = tempfile.mkdtemp(prefix="rf_isa_")
        files = build_asm(tmp)
    total, nk, nm = 0, 0, 0
    for f in files:
        for name, ins in parse(f).items():
            nk += 1
            nm += sum(1 for op, _, _ in ins if op.startswith("v_mfma"))
            for (k, need, ws, mf, use, kind) in scan_kernel(name, ins):
                total += 1
                print(f"{os.path.basename(f)}: {k}\n    {mf}\n    -> {use}   [{kind}: {ws} wait states, {need} wanted]")
    print(f"scanned {len(files)} files, {nk} kernels, {nm} MFMA instructions: {total} early reads of an MFMA result")
    return 1 if total else 0


if __name__ == "__main__":
    sys.exit(main())
