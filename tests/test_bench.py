"""bench.py end to end on the GPU box: the one-GPU line (two lanes, torch-free, both BASELINE configs, the streaming
leg), the RCCL exchange on a single-rank communicator, and the N = 2 path with the exchange over gloo (the box has one
GPU and RCCL refuses two ranks on one device)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SMALL = ["--steps", "6", "--warmup", "2", "--batch", "16", "--lanes", "2", "--templates", "300", "--no-cpu-baseline", "--latency-calls", "40"]


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
    assert rf["bound"] == "l2" and rf["achieved"] > 0 and rf["frames_per_launch"] == 8
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3
    assert rf["load_bytes_per_launch"] > 0 and rf["hbm_algorithmic"]["algorithmic_bytes_per_launch"] > 0
    h = d["config"]["h2d_inclusive"]
    assert h["value"] > 0 and h["h2d_GBps"] > 0 and h["matches_step0_step1"][0] > 0
    # the streaming leg matches different frames per slot every step
    assert d["config"]["baseline_config"] == 2
    # r05: the reference's own call pattern in the line (one 640x480 colour-only frame per call, 1950 templates) and the step's vector-issue roof
    lat = d["latency"]
    assert "error" not in lat, lat
    for leg in ("resident_frame", "pinned_host_frame", "pageable_host_frame"):
        assert 0 < lat[leg]["min_us"] <= lat[leg]["median_us"] <= lat[leg]["p95_us"] < 20000, lat
    assert lat["resident_frame"]["median_us"] <= lat["pageable_host_frame"]["median_us"]
    assert sum(lat["gpu_stage_us_resident"].values()) <= lat["resident_frame"]["median_us"] * 1.2
    assert "roofline_pipeline" in d and ("frac" in d["roofline_pipeline"])       # None + reason when no counter file matches this (small) command


@pytest.mark.gpu
def test_bench_config3_line():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", "3", "--no-h2d"] + SMALL,
                       capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    d = _json_line(r.stdout)
    assert d["config"]["baseline_config"] == 3 and "1280x960" in d["config"]["workload"] and d["value"] > 0
    # fixed geometry of SURVEY.md 8d config 3: 31 features x P = 3909 positions per template
    assert d["roofline"]["hbm_algorithmic"]["algorithmic_bytes_per_launch"] == 300 * 31 * 3909 * 8


@pytest.mark.gpu
@pytest.mark.parametrize("config,batch", [(2, 32), (3, 32)])
def test_bench_accepts_its_line_on_the_timed_buffers(config, batch):
    """r06 (VERDICT r5 #1b): cpu_baseline compares what the TIMED lanes wrote into their own result buffers in the last timed step (frames 0..3 of
    lane 0, the first frame of the other lane) with the oracle -- for config 3 that is the bit-plane scan on 16-frame launches -- and says so."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", str(config), "--steps", "4", "--warmup", "1", "--batch", str(batch),
                        "--lanes", "2", "--templates", "300", "--no-h2d", "--no-latency", "--cpu-seconds", "1"],
                       capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    d = _json_line(r.stdout)
    cb = d["cpu_baseline"]
    assert "error" not in cb, cb
    assert cb["timed_buffers_checked"] == [0, 1, 2, 3, batch // 2] and "TIMED lanes' own result buffers" in cb["sample"]
    assert cb["value"] > 0 and cb["kind"] == "port"
    if config == 3:
        assert d["roofline"]["kernel"] == "k_scan1" and "k_scan1" in cb["sample"]


@pytest.mark.gpu
def test_bench_config2_default_takes_the_lds_scan():
    """r06: config 2's lane-steps (640x480 RGB-D, 32 frames here) take the bit-plane scan with the planes in LDS by the default cost rule; the line names
    the kernel, prices it against the LDS read rate (no counter file matches this small command) and is accepted on the timed buffers."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", "2", "--steps", "4", "--warmup", "1", "--batch", "64",
                        "--lanes", "2", "--templates", "600", "--no-h2d", "--no-latency", "--cpu-seconds", "1"],
                       capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    d = _json_line(r.stdout)
    rf = d["roofline"]
    assert rf["kernel"] == "k_scanl" and rf["bound"] in ("lds", "valu") and rf["scan1_lanes_per_frame"] >= 1000 and 0 < rf["frac"] < 1
    assert "error" not in d["cpu_baseline"] and "k_scanl" in d["cpu_baseline"]["sample"]


@pytest.mark.gpu
def test_bench_rccl_single_rank_communicator():
    """The whole gathered path (k_pack_lists, 2 x ncclAllGather per lane-step, merge of the owned frames) through a
    single-rank RCCL communicator: what a 1-GPU box can run of the N > 1 path."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29541")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--force-rccl", "--no-h2d"] + SMALL,
                       capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    d = _json_line(r.stdout)
    assert d["n_gpus"] == 1 and d["value"] > 0 and d["config"]["matches_frame0"] > 0
    assert "ncclAllGather" in d["config"]["exchange"]


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


@pytest.mark.gpu
def test_bench_launcher_two_ranks_functional():
    """Plain `python bench.py --gpus 2` (no torchrun, WORLD_SIZE unset): the parent starts the two ranks itself and relays
    rank 0's line (here with the exchange over gloo, the box has one GPU)."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--functional-gloo"] + SMALL,
                       capture_output=True, text=True, timeout=1200, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    d = _json_line(r.stdout)
    assert d["n_gpus"] == 2 and d["config"]["templates_total"] == 600 and d["config"]["functional_gloo"] is True
    assert d["config"]["matches_frame0"] > 0


@pytest.mark.gpu
def test_bench_launcher_refuses_two_ranks_on_one_gpu():
    """On a 1-GPU box `bench.py --gpus 2` must fail loudly instead of printing a 1-GPU number labelled n_gpus: 2
    (VERDICT r2 #1): rank 1 finds no second device, the launcher stops the other rank, exit code non-zero, no JSON."""
    import importlib
    lm = importlib.import_module("line-mod-pipeline_amd")
    probe = lm.Detector(color_only=True, width=64, height=64, device=1)
    try:
        probe.prepare_slot(0)
    except lm.LinemodError as e:
        if "no frame" in str(e):
            pytest.skip("this box has a second GPU")
    finally:
        probe.close()
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"] + SMALL,
                       capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert r.returncode != 0
    assert not [l for l in r.stdout.splitlines() if l.startswith('{"metric"')]
    assert "rank" in r.stderr and "void" in r.stderr


def test_bench_launcher_without_gpu_fails_loudly():
    """CPU container: no HIP device -> every rank fails, the launcher reports it, no JSON line, non-zero exit."""
    import importlib
    lm = importlib.import_module("line-mod-pipeline_amd")
    probe = lm.Detector(color_only=True, width=64, height=64)
    try:
        probe.prepare_slot(0)
        pytest.skip("a HIP device is present")
    except lm.LinemodError as e:
        if e.code != lm.LM_ERR_NO_DEVICE:
            pytest.skip("a HIP device is present")
    finally:
        probe.close()
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"] + SMALL,
                       capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert r.returncode != 0
    assert not [l for l in r.stdout.splitlines() if l.startswith('{"metric"')]
    assert "void" in r.stderr


def test_bench_rejects_mismatched_world_size():
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4"] + SMALL,
                       capture_output=True, text=True, timeout=120, cwd=ROOT, env=env)
    assert r.returncode != 0 and "does not match WORLD_SIZE" in r.stderr
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + SMALL,       # --gpus defaults to 1
                       capture_output=True, text=True, timeout=120, cwd=ROOT, env=env)
    assert r.returncode != 0 and "does not match WORLD_SIZE" in r.stderr


@pytest.mark.gpu
def test_bench_config5_line():
    """bench.py --config 5: the hot path of BASELINE configs[4] -- 8-frame batches of 1280x960 RGB-D against three classes x
    8100 templates in one class-list match (bank fixed: reported as strong scaling)."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", "5", "--no-h2d", "--steps", "4", "--warmup", "1",
                        "--no-cpu-baseline"], capture_output=True, text=True, timeout=1200, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    d = _json_line(r.stdout)
    assert d["config"]["baseline_config"] == 5 and d["config"]["templates_total"] == 24300 and d["scaling"] == "strong"
    assert d["config"]["frames_per_step"] == 24 and d["config"]["lanes"] == 3 and d["value"] > 0       # three lanes x 8-frame batches
    assert d["config"]["matches_frame0"] > 0


def test_counter_files_are_used_only_for_the_command_they_were_collected_with(tmp_path):
    """the roofline figures that come from committed PMC passes must describe THIS run: bench.load_counters takes a
    file only when workload, launch shape, threshold, templates, scan variant and the kernel sources' hash all agree, and says
    which key differed otherwise.  preprocess_roofline turns a matching file into per-kernel fractions (CPU-only test)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    sha = bench.kernel_source_sha16()
    assert len(sha) == 16 and sha == bench.kernel_source_sha16()
    meta = {"baseline_config": 2, "frames_per_launch": 96, "threshold": 80.0, "templates_per_gpu": 3000, "scan_variant": 0,
            "scan_form": 0, "no_prune": False, "byte_responses": False, "kernel_source_sha16": sha}
    assert bench.load_counters(meta, str(tmp_path)) == (None, "no profiles/*_counters_c2.json committed")
    kernels = {
        "k_scan4<6, true, 2, false>": {"calls": 33, "avg_us": 167.0, "hbm_bytes_per_launch": 60e6, "TCP_TCC_READ_REQ_sum": 32.4e6,
                                       "TCC_HIT_sum": 98.6, "TCC_MISS_sum": 1.4, "GRBM_GUI_ACTIVE": 8 * 2.4e3 * 167.0, "GRBM_GUI_ACTIVE_instances": 1},
        "k_blur_pyr<16>": {"calls": 33, "avg_us": 70.0, "hbm_bytes_per_launch": 311e6, "hbm_read_bytes_per_launch": 200e6,
                           "hbm_write_bytes_per_launch": 111e6, "SQ_ACTIVE_INST_VALU": 0.5 * 1024 * (70.0 * 2.4e3) / 4, "SQ_INSTS_VALU": 3.0e7,
                           "GRBM_GUI_ACTIVE": 8 * 168000.0, "GRBM_GUI_ACTIVE_instances": 8},
        "k_refine_plan": {"calls": 33, "avg_us": 1.0},
    }
    for name, m in (("r04_counters_c2.json", dict(meta, threshold=60.0)), ("r03_counters_c2.json", dict(meta, kernel_source_sha16="0" * 16))):
        json.dump({"meta": m, "kernels": kernels, "bytes_per_request": 128}, open(tmp_path / name, "w"))
    got, why = bench.load_counters(meta, str(tmp_path))
    assert got is None and "threshold" in why and "kernel_source_sha16" in why
    json.dump({"meta": meta, "kernels": kernels, "bytes_per_request": 128}, open(tmp_path / "r05_counters_c2.json", "w"))
    got, why = bench.load_counters(meta, str(tmp_path))
    assert why is None and got["source"].endswith("r05_counters_c2.json")
    one_lane = {"stage_us": [380.0 * 10, 0, 0, 0], "launches": 10}
    pre = bench.preprocess_roofline(got, None, one_lane, 96)
    by = {e["kernel"]: e for e in pre["kernels"]}
    b = by["k_blur_pyr<16>"]
    assert b["stage"] == "preprocess" and abs(b["hbm_GBps"] - 311e6 / 70e-6 / 1e9) < 1 and abs(b["frac_of_hbm_peak"] - b["hbm_GBps"] / 8000) < 1e-3
    assert abs(b["valu_busy"] - 0.5) < 1e-3 and b["bound"] == "hbm"          # 4.4 TB/s = 0.70 of the achievable 6.3
    assert "k_refine_plan" not in by and by["k_scan4<6, true, 2, false>"]["stage"] == "match"
    assert pre["preprocess_hbm_bytes_per_frame"] == round(311e6 / 96) and pre["preprocess_live_us_per_launch"] == 380.0
    assert bench.preprocess_roofline(None, "because", one_lane, 96) == {"kernels": None, "reason": "because"}


@pytest.mark.gpu
def test_bench_frame_shard_on_a_single_rank_communicator():
    """--parallelism frame-shard (whole bank per rank, own frames per rank, no exchange): on a 1-GPU box the path runs with a
    single-rank RCCL communicator that only carries the barrier and the max over ranks; the line says so."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--parallelism", "frame-shard", "--force-rccl", "--no-h2d"] + SMALL,
                       capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    d = _json_line(r.stdout)
    assert d["config"]["parallelism"] == "frame-shard x1" and d["scaling"] == "weak" and d["value"] > 0
    assert d["config"]["templates_per_gpu"] == d["config"]["templates_total"] == 300
    assert "none in the data path" in d["config"]["exchange"] and d["config"]["rccl_ranks"] == 1
    assert d["config"]["matches_frame0"] > 0
