"""The committed evidence belongs to the committed kernels (CPU): the PMC summaries bench.py quotes `roofline.traffic` from carry
the hash of the csrc/ tree they were collected on, and the committed bench lines keep the driver's JSON contract."""
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


@pytest.mark.parametrize("name", ["r04_traffic_pmc.json", "r04_mfma_pmc.json", "r04_config4_traffic_pmc.json",
                                  "r04_config4_mfma_pmc.json", "r04_config5_traffic_pmc.json", "r04_config5_mfma_pmc.json"])
def test_pmc_files_belong_to_this_tree(name):
    with open(os.path.join(ROOT, "profiles", name)) as fh:
        pm = json.load(fh)
    assert pm["tree"] == bench.tree_hash(), f"profiles/{name} was collected on another csrc/ tree: run tools/collect_profiles.sh"
    assert pm["families"]


BASE_KEYS = {"metric": str, "value": float, "unit": str, "n_gpus": int, "steps": int, "warmup": int, "ms_per_step": float,
             "higher_is_better": bool, "scaling": str, "dtype": str, "data": str, "config": dict}


@pytest.mark.parametrize("name,full", [("r04_bench_default.json", True), ("r04_bench_config4.json", False),
                                       ("r04_bench_config5.json", False)])
def test_committed_bench_lines_keep_the_contract(name, full):
    with open(os.path.join(ROOT, "profiles", name)) as fh:
        d = json.loads(fh.read().strip().splitlines()[-1])
    for k, t in BASE_KEYS.items():
        assert isinstance(d[k], t), (k, type(d[k]))
    assert d["vs_baseline"] is None and "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s")
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    assert r["traffic"] is None or r["traffic"] > 0
    if full:
        assert r["traffic"] is not None, r.get("traffic_source")
        c = d["cpu_baseline"]
        assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["value"] > 0 and c["sample"]
        assert d["parity"]["modes"]["fp16"]["dist_argmax_agreement"] > 0.95
