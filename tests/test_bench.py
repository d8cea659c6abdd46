"""bench.py end to end on the GPU box: the one-GPU line (two lanes, torch-free) and the whole N = 2 path (rank
processes with torch.distributed + torch-free matcher workers, exchange over gloo because the box has one GPU)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SMALL = ["--steps", "6", "--warmup", "2", "--batch", "16", "--templates", "300", "--no-cpu-baseline"]


def _json_line(out):
    lines = [l for l in out.splitlines() if l.startswith('{"metric"')]
    assert len(lines) == 1, out[-3000:]
    return json.loads(lines[0])


@pytest.mark.gpu
def test_bench_one_gpu_line():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + SMALL, capture_output=True, text=True,
                       timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    d = _json_line(r.stdout)
    assert d["metric"] == "detections/sec" and d["n_gpus"] == 1 and d["steps"] == 6 and d["value"] > 0
    assert d["config"]["lanes"] == 2 and d["config"]["frames_per_step"] == 16
    rf = d["roofline"]
    assert rf["bound"] == "hbm" and rf["achieved"] > 0 and rf["frames_per_launch"] == 8
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3
    assert rf["on_chip"]["load_bytes_per_launch"] > 0


@pytest.mark.gpu
def test_bench_two_ranks_functional():
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", "29533", os.path.join(ROOT, "bench.py"), "--gpus", "2",
           "--functional-gloo"] + SMALL
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=1200, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    d = _json_line(r.stdout)
    assert d["n_gpus"] == 2 and d["config"]["templates_total"] == 600 and d["value"] > 0
    assert d["config"]["matches_frame0"] > 0          # merged list of both shards for frame 0
