"""CPU-side ISA audit of the shipped kernels (no GPU): tools/isa_hazard_scan.py compiles every csrc/*.hip to gfx950 assembly
(both 16-bit builds) and checks that no instruction reads an MFMA result before its wait states have passed, on fall-through
paths AND across branches -- the defect behind round 2's non-reproducible FAVOR+ kernel (hipcc inserted no s_nop on the
loop-exit path between an accumulator's last v_mfma and the ds_bpermute_b32 that broadcast it)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import isa_hazard_scan as S  # noqa: E402

BAD = """
_Zfake_kernel:
.LBB0_1:
	v_mfma_f32_16x16x32_bf16 v[26:29], v[30:33], v[6:9], v[26:29]
	s_cbranch_vccz .LBB0_2
	s_nop 7
	v_add_f32_e32 v1, v26, v26
	s_branch .LBB0_1
.LBB0_2:
	ds_bpermute_b32 v9, v54, v26
	s_endpgm
"""


def test_scanner_flags_the_round2_pattern(tmp_path):
    """the exact shape of the defect: last MFMA of the accumulator, taken branch, LDS read of the result -- and the
    fall-through path with its s_nop is accepted"""
    f = tmp_path / "bad.s"
    f.write_text(BAD)
    k = S.parse(str(f))
    found = S.scan_kernel("_Zfake_kernel", k["_Zfake_kernel"])
    assert len(found) == 1 and "ds_bpermute_b32" in found[0][4] and found[0][2] < 8


def test_no_shipped_kernel_reads_an_mfma_result_early():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "isa_hazard_scan.py")], capture_output=True, text=True, timeout=900)
    tail = r.stdout.strip().splitlines()[-1]
    print(tail)
    assert r.returncode == 0 and tail.endswith(": 0 early reads of an MFMA result"), r.stdout[-3000:]
    assert "kernels" in tail and int(tail.split(" kernels, ")[1].split(" ")[0]) > 10000  # both builds really were scanned
