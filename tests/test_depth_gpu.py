"""Whole-model parity AT DEPTH against the CPU oracle (missing #2): config-2 dimensions (N=128, L=256,
d_msa=384, d_pair=288, 12/8/4 heads, k=128/32), B=1, 2 two-track + 2 three-track(+final) blocks of 4 encoder layers each
= 4 of the 13 blocks of the benchmarked model, every compute mode of the library against oracle.rosettafold_forward on
the same seeded weights and inputs (one oracle forward, ~60-90 s of CPU, shared by the three modes).

Asserted (relative L2 over a whole tensor; distogram argmax agreement over all L x L pairs and over the pairs whose top-2
oracle logits differ by more than 2 % of the map's range, "clear margin"):

  mode   logits rel-L2   xyz rel-L2   argmax all pairs   argmax clear margin    observed (round 3, profiles/r03_depth_parity_oracle.json)
  fp32   < 5e-4          < 5e-4       == 1.0 (distogram) == 1.0                 5.8e-6 / 9.3e-6 / 1.0 / 1.0
  fp16   < 1e-2          < 0.15       >= 0.99            >= 0.999               3.4e-3 / 0.051  / 0.9958 / 1.0
  bf16   < 6e-2          < 0.3        >= 0.93            >= 0.99                3.0e-2 / 0.114  / 0.9645 / 1.0
(fp32: exact fp32 tiles, the strict claim; fp16: 16-bit MFMA at the full rate, the parity mode at speed; bf16: the dtype
BASELINE.json quotes the metric on.  The other three maps (theta / phi / omega) must agree on >= 0.9999 of the pairs in
fp32 mode: at random init single pairs are exact ties to the last bit.)

tools/depth_parity.py prints the block-by-block error curves behind these numbers (profiles/r03_depth_parity*.json).
"""
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
if not torch.cuda.is_available():
    pytest.skip("needs a GPU", allow_module_level=True)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import depth_parity as DP  # noqa: E402


class _Args:
    oracle, struct_lowp, modes, B, N, L, n_two, n_three = True, False, "fp32,fp16,bf16", 1, 128, 256, 2, 2


@pytest.fixture(scope="module")
def result():
    return DP.run(_Args())


def test_fp32_mode_matches_the_oracle_at_depth(result):
    r = result["fp32"]
    print("\n[depth fp32]", r["rel_l2"], r["dist_argmax_agreement"])
    assert all(v < 5e-4 for v in r["rel_l2"].values()), r["rel_l2"]
    assert r["dist_argmax_agreement"] == 1.0 and r["dist_argmax_agreement_clear_margin"] == 1.0
    assert all(v >= 0.9999 for v in r["argmax_agreement"].values()), r["argmax_agreement"]


def test_fp16_mode_meets_the_parity_bar_at_depth(result):
    r = result["fp16"]
    print("\n[depth fp16]", r["rel_l2"], r["dist_argmax_agreement"], r["dist_argmax_agreement_clear_margin"])
    for k in ("theta", "phi", "dist", "omega"):
        assert r["rel_l2"][k] < 1e-2, (k, r["rel_l2"])
    assert r["rel_l2"]["xyz"] < 0.15
    assert r["dist_argmax_agreement"] >= 0.99
    assert r["dist_argmax_agreement_clear_margin"] >= 0.999


def test_bf16_mode_bound_at_depth(result):
    r = result["bf16"]
    print("\n[depth bf16]", r["rel_l2"], r["dist_argmax_agreement"], r["dist_argmax_agreement_clear_margin"])
    for k in ("theta", "phi", "dist", "omega"):
        assert r["rel_l2"][k] < 6e-2, (k, r["rel_l2"])
    assert r["rel_l2"]["xyz"] < 0.3
    assert r["dist_argmax_agreement"] >= 0.93
    assert r["dist_argmax_agreement_clear_margin"] >= 0.99


def test_error_grows_monotonically_enough(result):
    """the per-block curve is the evidence that the gap is rounding compounding through depth, not a defect in one block:
    no block multiplies the pair-stream error of the 16-bit modes by more than 4x"""
    for mode in ("fp16", "bf16"):
        pair = [row["pair"] for row in result[mode]["curve"] if "pair" in row]
        print(f"\n[depth {mode}] pair rel-L2 by block: {[round(v, 5) for v in pair]}")
        for a, b in zip(pair, pair[1:]):
            assert b < 4 * a + 1e-3, (mode, pair)


# ---- the benchmarked depth (8 two-track + 4 three-track + final block), round 4 ------------------------------------------------
# One oracle forward at this depth takes 3.5-5 min of CPU: it runs as a tool (tools/depth_parity.py --oracle --full ->
# profiles/r04_depth_parity_oracle_full.json: fp32 mode vs oracle logits 6.7e-5 / distogram argmax 0.99995 / xyz 1.1e-2 -- even
# two fp32 implementations that agree to 6e-7 on the streams drift apart in xyz through the five structure blocks).  The test
# below asserts the 16-bit modes at the full depth against the library's exact-fp32 mode, with bounds placed beside the
# yardstick profiles/r04_oracle_sensitivity.json: the CPU ORACLE against itself with nothing changed but its WEIGHTS rounded to
# the 16-bit type (fp16: logits 5e-3, argmax 0.9934, xyz 0.178; bf16: 2.2e-2, 0.971, 0.298).  xyz of a 16-bit mode cannot be
# better than that (measured: fp16 0.21, bf16 0.35).  The logits were 4-7x above the yardstick until the operands that carry
# a per-sample constant were conditioned (csrc/condition.hip; tools/precision_probe.py --module-sweep / --pum-sweep /
# --gemm-sweep found them: PairUpdateWithMsa's tiled 1-D features and first convolution, the head's LayerNorm operand, the
# attention layers' value path): fp16 2.0e-2 -> 4.9e-3 (argmax 0.9725 -> 0.9940), bf16 0.157 -> 0.026 (0.802 -> 0.9655) -- AT the
# yardstick; against the ORACLE at this depth (profiles/r04_depth_parity_oracle_full.json): fp16 5.5e-3 / 0.9931, bf16 2.7e-2 /
# 0.9639.  DESIGN.md section 4.
class _ArgsFull:
    oracle, full, struct_lowp, modes, B, N, L, n_two, n_three = False, False, False, "fp16,bf16", 1, 128, 256, 8, 5


@pytest.fixture(scope="module")
def result_full():
    return DP.run(_ArgsFull())


def test_fp16_mode_at_the_benchmarked_depth(result_full):
    r = result_full["fp16"]
    print("\n[full depth fp16]", r["rel_l2"], r["dist_argmax_agreement"], r["dist_argmax_agreement_clear_margin"])
    for k in ("theta", "phi", "dist", "omega"):
        assert r["rel_l2"][k] < 8e-3, (k, r["rel_l2"])                 # observed 4.5-5.1e-3 (before the conditioning: 1.8-2.1e-2)
    assert r["dist_argmax_agreement"] >= 0.99                           # observed 0.9940 (0.973)
    assert r["dist_argmax_agreement_clear_margin"] >= 0.9995            # observed 1.0
    assert r["rel_l2"]["xyz"] < 2 * 0.178                               # twice the oracle's own weight-rounding drift; observed 0.21
    pair = [row["pair"] for row in r["curve"] if "pair" in row]
    assert max(pair) < 1.5e-3, pair                                     # the residual streams stay within 0.1 % (observed 9.6e-4)


def test_bf16_mode_at_the_benchmarked_depth(result_full):
    r = result_full["bf16"]
    print("\n[full depth bf16]", r["rel_l2"], r["dist_argmax_agreement"], r["dist_argmax_agreement_clear_margin"])
    for k in ("theta", "phi", "dist", "omega"):
        assert r["rel_l2"][k] < 0.04, (k, r["rel_l2"])                  # observed 0.023-0.027 (before the conditioning: 0.14-0.16)
    assert r["dist_argmax_agreement"] >= 0.95                           # observed 0.9655 (0.80)
    assert r["dist_argmax_agreement_clear_margin"] >= 0.995             # observed 1.0 (0.983-0.993)
    assert r["rel_l2"]["xyz"] < 2 * 0.298                               # observed 0.346
    pair = [row["pair"] for row in r["curve"] if "pair" in row]
    assert max(pair) < 1.2e-2, pair                                     # observed 7.3e-3
