#!/usr/bin/env python3
"""bench.py -- LINE-MOD hot path on MI355X (BASELINE.json metric: detections/sec at 640x480 RGB-D,
~3000 templates, 2-level pyramid, ColorGradient+DepthNormal).

A "step" = one pass of the hot path (Detector::match, a3-a15) over one batch of `--batch` synthetic
frames that are already resident in HBM.  One process per GPU; with N > 1 the template bank is
sharded over the ranks (3000 templates per GPU, weak scaling), every rank matches the same frames
against its shard and the per-shard match lists are exchanged with one RCCL all-gather per step
and merged on every rank.

value = n_gpus * frames / time: one detection = one frame searched against one 3000-template shard
(frames/sec of the whole job is reported separately in config.frames_per_sec).

Prints ONE JSON line on rank 0 (see the driver contract in the task description) carrying
`roofline` (similarity-scan kernel, HIP events on its launch stream over the timed region) and
`cpu_baseline` (the CPU oracle timed on this host, N=1 only).
"""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

L2_PEAK_GBS = 34500.0     # MI355X_MICROARCH.md, "L2 (per XCD)": ~34.5 TB/s aggregate
HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def quantized_from_gpu(det, bgr, depth, M, L=2):
    """Quantised images of a frame from the product's own kernels (used to cut crop templates)."""
    det.upload_frame(0, bgr, depth if M == 2 else None)
    det.prepare_slot(0)
    return {(l, m): det.debug_read(0, 0, l, m).reshape(det.height >> l, det.width >> l)
            for l in range(L) for m in range(M)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=256, help="frames per step (resident in HBM)")
    ap.add_argument("--lanes", type=int, default=0, choices=(0, 1, 2),
                    help="2: the step's frames are split over the detector's two lanes (two HIP streams driven by one "
                         "host thread through lm_match_begin / lm_match_end) so that the stages of one half overlap "
                         "those of the other (the scan is L1/L2-bound, the preprocess passes VALU / fabric-bound); "
                         "0 = 2 on one GPU, 1 with N > 1 (see the note on torch in main())")
    ap.add_argument("--templates", type=int, default=3000, help="templates per GPU")
    ap.add_argument("--threshold", type=float, default=80.0)
    ap.add_argument("--byte-responses", action="store_true", help="LM_FLAG_BYTE_RESPONSES: byte scan kernel (A/B)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--functional-gloo", action="store_true",
                    help="functional check of the N > 1 path on a 1-GPU box: every rank uses cuda:0 and the exchange "
                         "goes through gloo on the host (never a measurement)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the cpu_baseline sample")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d does not match WORLD_SIZE %d" % (args.gpus, world))

    # One GPU needs no torch at all (lm_match_end synchronises its stream; inputs and results are numpy), and it is
    # kept out on purpose: with torch loaded into the process the detector's two HIP streams no longer overlap
    # (measured on the MI355X box, 100 steps of 128 frames: 93.8 K detections/s without `import torch`, 76 K with
    # it, whatever the initialisation order, thread counts or synchronisation calls; one lane is unaffected:
    # 85 K both ways).  N > 1 needs torch.distributed, so there the default is one lane.
    torch = dist = device = None
    if world > 1:
        import torch
        import torch.distributed as dist
        if args.functional_gloo:
            local_rank = 0
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        if args.functional_gloo:
            dist.init_process_group(backend="gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group(backend="nccl", rank=rank, world_size=world,
                                    device_id=torch.device("cuda", local_rank))
        device = torch.device("cuda", local_rank)
        torch.cuda.synchronize()

    lm = importlib.import_module("line-mod-pipeline_amd")
    synth = importlib.import_module("line-mod-pipeline_amd.synth")
    distmod = importlib.import_module("line-mod-pipeline_amd.dist")

    W, H, M, B = 640, 480, 2, args.batch
    n_total = args.templates * world
    cfg = lm.default_config(color_only=False, width=W, height=H, device=local_rank, shard_rank=rank,
                            shard_size=world, frame_slots=max(B, 1), flags=1 if args.byte_responses else 0)
    det = lm.Detector(cfg)

    # ---- workload: seeded synthetic frames + fixed-geometry bank
    frames = [synth.make_frame(W, H, seed=1234 + i) for i in range(B)]
    q = quantized_from_gpu(det, frames[0][0], frames[0][1], M)
    descs, feats, crops = synth.make_bank(n_total, M, 2, seed=4321, fixed_l0_size=(96, 96), quantized=q,
                                          crop_fraction=0.1, frame_size=(W, H), T0=det.get_T(0))
    det.add_class("synthetic.ply", descs, feats)
    for i, (bgr, depth) in enumerate(frames):
        det.upload_frame(i, bgr, depth)

    cap = 4096
    NL = args.lanes if args.lanes else (2 if world == 1 else 1)
    if B % 2 or B < 2:
        NL = 1
    Bl = B // NL                                         # frames per lane and launch
    NBUF = 3                                             # result buffers in rotation
    bufs = [(np.zeros((B, cap), lm.MATCH_DTYPE), np.zeros(B, np.int32)) for _ in range(NBUF)]
    views = [[(o[l * Bl:(l + 1) * Bl], cn[l * Bl:(l + 1) * Bl]) for l in range(NL)] for o, cn in bufs]
    gather = None
    if world > 1:
        import queue
        import threading
        gather = distmod.ShardGather(lm.merge_matches, cap=cap, pack_fn=lm.pack_matches, merge_batch_fn=lm.merge_batch,
                                     device=torch.device("cpu") if args.functional_gloo else device)
    state = {"k": 0, "merged": None, "error": None}

    def run_steps(n):
        """n passes of the hot path over the batch, driven by this one host thread: lane l works on the frames of
        slots [l * Bl, (l + 1) * Bl); as soon as a lane's step is collected its next step is enqueued, so the two
        streams always have work and their stages overlap.  With N > 1 an exchange thread gathers + merges step k
        (two small collectives + lm_merge_batch) while the GPU already works on step k + 1."""
        if n <= 0:
            return
        k0 = state["k"]
        state["k"] += n
        if gather is not None:
            todo = queue.Queue()
            free = threading.Semaphore(NBUF - 1)           # buffers the lanes may fill ahead of the exchange

            def exchange():
                while True:
                    k = todo.get()
                    if k is None:
                        return
                    try:
                        if state["error"] is None:
                            state["merged"] = gather.gather_merge(*bufs[k % NBUF])
                    except Exception as e:                 # noqa: BLE001 - re-raised by the main thread
                        state["error"] = e
                    free.release()

            th = threading.Thread(target=exchange)
            th.start()
        for l in range(NL):
            det.match_begin(l, l * Bl, Bl, args.threshold, 0)
        for k in range(k0, k0 + n):
            if gather is not None:
                free.acquire()
            for l in range(NL):
                o, cn = views[k % NBUF][l]
                det.match_end(l, cap, out=o, counts=cn)
                if k + 1 < k0 + n:
                    det.match_begin(l, l * Bl, Bl, args.threshold, 0)
            if gather is not None:
                todo.put(k)
        if gather is not None:
            todo.put(None)
            th.join()
            if state["error"] is not None:
                raise state["error"]

    def fence():
        # N > 1: barrier + torch.cuda.synchronize().  N = 1: the same device-wide synchronisation without torch.
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()
        else:
            det.synchronize()                           # hipDeviceSynchronize: the torch-free torch.cuda.synchronize()

    run_steps(args.warmup)
    det.set_profiling(True)
    fence()
    t0 = time.perf_counter()
    run_steps(args.steps)
    fence()
    dt = time.perf_counter() - t0
    merged = state["merged"]
    prof = det.get_profile()
    det.set_profiling(False)
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=torch.device("cpu") if args.functional_gloo else device)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    frames_done = B * args.steps
    fps = frames_done / dt
    n_matches0 = int(bufs[(state["k"] - 1) % NBUF][1][0]) if merged is None else len(merged[0])

    # ---- roofline of the dominant kernel (similarity scan): algorithmic bytes / HIP-event time
    scan_us = prof["stage_us"][1] / max(prof["launches"], 1)
    bytes_per_launch = prof["scan_bytes"] / max(prof["launches"], 1)
    achieved = bytes_per_launch / (scan_us * 1e-6) / 1e9 if scan_us > 0 else 0.0
    traffic, traffic_src = pmc_traffic(Bl)
    l2_bytes = det.scan_load_bytes(0) * Bl         # bytes the scan's vector loads request per launch
    l2_rate = l2_bytes / (scan_us * 1e-6) / 1e9 if scan_us > 0 else 0.0
    roofline = {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_src,
                "kernel": "k_scan4" if not args.byte_responses else "k_scan", "avg_launch_us": round(scan_us, 2),
                "algorithmic_bytes_per_launch": bytes_per_launch, "frames_per_launch": Bl,
                "on_chip": {"bound": "l2", "load_bytes_per_launch": l2_bytes, "achieved": round(l2_rate, 1),
                            "peak": L2_PEAK_GBS, "unit": "GB/s", "frac": round(l2_rate / L2_PEAK_GBS, 4)},
                "note": "the scanned level's linear memories (0.6 MB/frame, nibble-packed) are L2-resident: the "
                        "algorithmic bytes are served by the L2s (on_chip: what the vector loads really request vs "
                        "the guide's 34.5 TB/s aggregate L2 rate); HBM traffic is ~1% of them (see DESIGN.md)"}
    stage_us_per_frame = [round(v / max(prof["frames"], 1), 2) for v in prof["stage_us"]]

    result = None
    if rank == 0:
        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            cpu = cpu_baseline(args, frames, descs, feats, det, lm)
        result = {
            "metric": "detections/sec", "value": round(world * fps, 1), "unit": "detections/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": "SURVEY 8d config 2: 640x480 RGB-D, ColorGradient+DepthNormal, T={5,8}, "
                                   "2-level pyramid, fixed-geometry 96x96 templates, threshold %g" % args.threshold,
                       "templates_per_gpu": args.templates, "templates_total": n_total, "frames_per_step": B,
                       "lanes": NL,
                       "frames_per_sec": round(fps, 1), "matches_frame0": n_matches0,
                       "unit_definition": "one detection = one frame matched against one %d-template bank shard; "
                                          "N GPUs search N shards of the same frames" % args.templates,
                       "stage_us_per_frame": dict(zip(["preprocess", "scan", "refine", "sort"], stage_us_per_frame)),
                       "stage_note": "HIP-event spans on each lane's stream per frame of that lane; with two lanes the "
                                     "spans of one lane contain kernels of the other, so they add up to more than the "
                                     "wall time per frame",
                       "parallelism": "template-shard x%d" % world},
            "roofline": roofline,
            "cpu_baseline": cpu,
        }
        print(json.dumps(result))
    det.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return result


def pmc_traffic(frames_per_launch):
    """HBM bytes per k_scan launch from the committed rocprofv3 PMC passes of this same command
    (profiles/summarize_pmc.py: 2 x FETCH_SIZE + WRITE_SIZE, KB -> bytes).  PMC counters cannot be read
    from inside the process, so this is null unless a committed pass matches frames_per_launch."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc_batch*.json")), reverse=True):
        try:
            d = json.load(open(path))
        except Exception:
            continue
        if d.get("frames_per_launch") != frames_per_launch:
            continue
        for k, v in d.get("kernels", {}).items():
            if k.startswith("k_scan"):
                return v["hbm_bytes_per_launch"], os.path.relpath(path, ROOT)
    return None, None


def cpu_baseline(args, frames, descs, feats, det, lm):
    """The CPU oracle (kind "port": a restatement, the reference itself cannot be built here) on this
    host's cores, bounded sample, same frames and bank; its match list must equal the GPU's first."""
    try:
        from oracle import oracle as O
        lib = O.build(arch="-march=native")
        cores = os.cpu_count() or 1
        orc = O.Detector(color_only=False, lib_path=lib)
        orc.add_class("synthetic.ply", descs, feats)
        bgr, depth = frames[0]
        gpu = det.match_slot(0, args.threshold, 0)
        exp = orc.match(bgr, depth, args.threshold, 0, threads=min(cores, 16))
        if gpu.tobytes() != exp.tobytes():
            return {"error": "GPU match list differs from the oracle: timing not accepted"}
        # all logical CPUs is not always fastest (SMT / cgroup limits): take the best of a few team sizes
        best_t, threads = None, cores
        for th in sorted({cores, max(cores // 2, 1), max(cores // 4, 1)}):
            t1 = time.perf_counter()
            orc.match(bgr, depth, args.threshold, 0, threads=th)
            t = time.perf_counter() - t1
            if best_t is None or t < best_t:
                best_t, threads = t, th
        t0 = time.perf_counter()
        n = 0
        while True:
            b, d = frames[n % len(frames)]
            orc.match(b, d, args.threshold, 0, threads=threads)
            n += 1
            if time.perf_counter() - t0 >= args.cpu_seconds or n >= 5000:
                break
        dt = time.perf_counter() - t0
        t1 = time.perf_counter()
        orc.match(bgr, depth, args.threshold, 0, threads=1)
        single = time.perf_counter() - t1
        return {"value": round(n / dt, 3), "unit": "detections/s", "cores": threads, "kind": "port",
                "sample": "%d full frames (a3-a15, same bank of %d templates) in %.1f s; OpenMP over templates and "
                          "over image rows, %d threads (fastest of {1/4, 1/2, all} of %d logical CPUs); upstream-"
                          "faithful single-thread run: %.3f s/frame; GPU and CPU match lists identical" %
                          (n, args.templates, dt, threads, cores, single)}
    except Exception as e:  # the bench line must still be printed
        return {"error": "%s: %s" % (type(e).__name__, e)}


if __name__ == "__main__":
    main()
