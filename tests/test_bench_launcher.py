"""`python bench.py --gpus N` must start N ranks itself (round-3 review: --gpus was parsed and never read, so the driver's
plain command would have recorded N = 1).  CPU only: --launch-check stops after the rendezvous (gloo), before any GPU call."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    return env


def test_gpus_2_starts_two_ranks():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--launch-check"], env=_env(),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout  # ONE JSON line, from rank 0
    js = json.loads(lines[0])
    assert js["n_gpus"] == 2 and js["ranks_seen"] == 2


def test_gpus_mismatch_is_an_error():
    env = dict(_env(), WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--launch-check"], env=env,
                       capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "WORLD_SIZE" in r.stderr


def test_child_exit_code_is_relayed():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--launch-check", "--config", "99"],
                       env=_env(), capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
