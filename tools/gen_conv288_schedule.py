#!/usr/bin/env python3
"""Generates the straight-line body of one super-step of csrc/conv288.hip: 39 fragment reads (inline asm) issued LOOKAHEAD
fragments ahead of the 108 MFMAs that consume them, with hand-counted lgkmcnt waits (LDS operations of a wave complete in
order).  The text between the GENERATED markers of conv288.hip is this script's output:  python tools/gen_conv288_schedule.py"""
WM, WN, LOOK, RING = 4, 9, 6, 8
DMA_AT = []   # (spreading the DMA pieces between the MFMAs was measured slower: 367 vs 353 us -- they are issued in one burst behind the barrier)
stream = []  # ("a", t, ii) | ("b", t, j)
for t in range(3):
    stream += [("a", t, ii) for ii in range(WM)] + [("b", t, j) for j in range(WN)]
pos = {f: k for k, f in enumerate(stream)}
bcount = {}
n = 0
for f in stream:
    if f[0] == "b":
        bcount[f] = n
        n += 1

def issue(f):
    if f[0] == "a":
        _, t, ii = f
        return f"CONV_RD(af[{t & 1}][{ii}], a_rd{ii} + sbase + {t} * dstep, 0);" if False else f"CONV_RDA(af[{t & 1}][{ii}], {ii}, {t});"
    _, t, j = f
    return f"CONV_RD(bq[{bcount[f] % RING}], b_rd[{j}] + sbase, {t} * B_BYTES);"

out = []
issued = 0
def issue_until(k):
    global issued
    while issued < min(len(stream), k):
        out.append("    " + issue(stream[issued]))
        issued += 1
issue_until(WM + LOOK)
for t in range(3):
    for j in range(WN):
        need = pos[("b", t, j)]
        allowed = issued - 1 - need
        regs = [f"bq[{bcount[('b', t, j)] % RING}]"] + ([f"af[{t & 1}][{ii}]" for ii in range(WM)] if j == 0 else [])
        ops = ", ".join(f'"+v"({r})' for r in regs)
        out.append(f'    asm volatile("s_waitcnt lgkmcnt({allowed})" : {ops});')
        for ii in range(WM):
            out.append(f"    acc[{ii}][{j}] = rf_mfma16(bq[{bcount[('b', t, j)] % RING}], af[{t & 1}][{ii}], acc[{ii}][{j}], 0, 0, 0);")
        issue_until(need + 1 + LOOK + (WM if j >= WN - 3 else 0))   # the next tap's four pixel fragments ride in early
        step_no = t * WN + j
        if step_no in DMA_AT:                                        # the next super-step's DMA pieces, spread over the first half
            out.append(f"    CONV_DMA({DMA_AT.index(step_no)});")
        out.append("    __builtin_amdgcn_sched_barrier(0);")
assert issued == len(stream)
# ring safety: a b fragment's slot may be reused only after its MFMAs were issued
for f, c in bcount.items():
    for g_, c2 in bcount.items():
        if c2 == c + RING:
            # g_ is issued when issued index reaches pos[g_]; that happens after step of f iff pos[g_] - (LOOK + WM) > pos[f] roughly
            assert pos[g_] - pos[f] >= RING, (f, g_)
print("\n".join(out))
