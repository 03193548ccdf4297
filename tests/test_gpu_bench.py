"""bench.py itself on the GPU box: the N > 1 path (one process per rank, row gather every step, max-over-ranks timing, rank 0's JSON
line) must not rot before its first real 8-GPU run.  Two ranks share the one card of the box under SOFTSPOKEN_DIST_BACKEND=gloo
(RCCL wants one device per rank); the device work per rank is the single-GPU step."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(*args, env=None, timeout=900):
    e = dict(os.environ)
    e.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    e.update(env or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], cwd=ROOT, env=e, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, (r.stdout[-800:], r.stderr[-3000:])
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]                   # ONE JSON line, from rank 0
    return json.loads(lines[0])


def test_bench_two_ranks_on_one_card(build_all):
    common = ["--files", "4", "--steps", "1", "--warmup", "1", "--no-cpu-baseline", "--no-secondary"]
    one = _bench("--gpus", "1", *common)
    assert one["n_gpus"] == 1 and one["config"]["rccl_world_size"] == 1 and one["value"] > 0 and one["rows_last_step"] > 0
    assert one["roofline"]["frac"] > 0 and one["roofline"]["kernel"].startswith("conv3x3")
    assert one["roofline"]["mfma"]["issued_tflops"] >= one["roofline"]["mfma"]["achieved_tflops_algorithmic"]
    # the traffic figure is either measured for exactly this kernel at this duration, or null with the reason beside it
    ts = one["roofline"]["traffic_source"]
    assert (one["roofline"]["traffic"] is None) == ("traffic_null_because" in ts)
    two = _bench("--gpus", "2", *common, env={"SOFTSPOKEN_DIST_BACKEND": "gloo"})
    assert two["n_gpus"] == 2 and two["config"]["rccl_world_size"] == 2 and two["config"]["backend"] == "gloo"
    assert two["scaling"] == "weak" and two["steps"] == 1 and two["warmup"] == 1
    assert two["rows_last_step"] == 2 * one["rows_last_step"]  # every rank's rows reach rank 0 (weak scaling: each has its own 4 files)
    assert two["value"] > 0 and two["config"]["windows_per_step_per_gpu"] == one["config"]["windows_per_step_per_gpu"]
