#!/usr/bin/env python3
"""bench.py -- LINE-MOD hot path on MI355X (BASELINE.json metric: detections/sec at 640x480 RGB-D, ~3000 templates,
2-level pyramid, ColorGradient+DepthNormal = SURVEY.md 8d config 2; --config 3 = 1280x960 colour-only T = {2, 8}).

A "step" = one pass of the hot path (Detector::match, a3-a15) over one batch of `--batch` synthetic frames that are
already resident in HBM.  One process per GPU, no torch anywhere: with N > 1 every rank holds one contiguous
template_id shard of the bank (lm_config.shard_rank / shard_size), matches the same frames against it, and the per-shard
lists are exchanged by two ncclAllGather calls per lane-step issued from the C++ library on the lane's own stream
(lm_match_begin_gathered / lm_match_end_gathered); every rank merges the frames it owns.

value = frames per second answered against the WHOLE bank (all N shards).  --scaling weak (default): 3000 templates
per GPU, the bank grows with N (N = 8: 24 000 templates ~ BASELINE config 4); --scaling strong: the 3000-template bank
of config 2 split N ways.

Prints ONE JSON line on rank 0 carrying `roofline` (similarity-scan kernel) and `cpu_baseline` (the CPU oracle timed
on this host, N = 1 only) and, at N = 1, `config.h2d_inclusive`: the same path with fresh host frames uploaded every
step (SURVEY.md 8d's "H2D of frame -> D2H of matches" definition).
"""
import argparse
import importlib
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

L2_PEAK_GBS = 34500.0     # MI355X_MICROARCH.md, "L2 (per XCD)": ~34.5 TB/s aggregate
HBM_PEAK_GBS = 8000.0     # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
LDS_B32_PEAK_GBS = 75000.0  # MI355X_MICROARCH.md, "LDS": aggregate rate of ds_read_b32-sized reads with every CU streaming (~75 TB/s; b64 / b128: ~150 TB/s)
PCIE_PEAK_GBS = 63.0      # MI355X_MICROARCH.md, "Host link": PCIe Gen5 x16, 63 GB/s (spec)

CONFIGS = {
    2: dict(name="SURVEY 8d config 2: 640x480 RGB-D, ColorGradient+DepthNormal, T={5,8}, 2-level pyramid, "
                 "fixed-geometry 96x96 templates (level-1 bbox 48x48, P = 995)",
            W=640, H=480, color_only=False, l0_size=(96, 96), seed_frames=1234, seed_bank=4321, lanes=3, batch=288, h2d_group=24),
    3: dict(name="SURVEY 8d config 3: 1280x960, ColorGradient only, T={2,8}, 2-level pyramid, fixed-geometry 192x192 "
                 "templates (level-1 bbox 96x96, P = 3909)",
            W=1280, H=960, color_only=True, l0_size=(192, 192), seed_frames=2234, seed_bank=77, lanes=2, batch=256, h2d_group=16),
    # BASELINE configs[4] on the hot path: a batch of 8 frames of 1280x960 RGB-D against three classes x 8 100 templates
    # (162 viewpoints x 5 radii x 10 rotations each) in ONE class-list match (Detector::match(..., class_ids)): one
    # pre-processing per frame for the three classes.  The bank is fixed (24 300 templates; N ranks shard every class N
    # ways), so this line is NOT a weak-scaling line.  The PoseDetection post-processing of config 5 is host C++
    # (tests/test_facade.py), not part of the timed hot path.
    5: dict(name="BASELINE configs[4] / SURVEY 8d config 5 (hot path): batches of 8 frames of 1280x960 RGB-D, T={5,8}, three classes x "
                 "8100 templates (variable geometry, level-0 bbox 96..320) in one class-list match",
            W=1280, H=960, color_only=False, l0_size=None, size_range=(96, 320), seed_frames=1234, seed_bank=500, lanes=3, batch=24, h2d_group=8,
            classes=3, templates_per_class=8100),
}


def quantized_from_gpu(det, bgr, depth, M, L=2):
    """Quantised images of a frame from the product's own kernels (used to cut crop templates)."""
    det.upload_frame(0, bgr, depth if M == 2 else None)
    det.prepare_slot(0)
    return {(l, m): det.debug_read(0, 0, l, m).reshape(det.height >> l, det.width >> l)
            for l in range(L) for m in range(M)}


CAP = 4096       # match records per frame in the result buffers (a shard's list must fit: SURVEY.md 8e); Runner.cap grows with the bank
NBUF = 3         # result buffers in rotation


class Runner:
    """The matcher of one GPU: detector, resident synthetic workload, and the step loop over its lanes."""

    def __init__(self, args, rank, world, local_rank, exchange):
        self.lm = lm = importlib.import_module("line-mod-pipeline_amd")
        synth = importlib.import_module("line-mod-pipeline_amd.synth")
        self.args, self.rank, self.world, self.exchange = args, rank, world, exchange
        wl = CONFIGS[args.config]
        W, H, B = wl["W"], wl["H"], args.batch
        self.M = M = 1 if wl["color_only"] else 2
        self.B, self.W, self.H = B, W, H
        self.n_total = args.templates_total
        flags = (lm.FLAG_BYTE_RESPONSES if args.byte_responses else 0)
        quota = cgroup_cpus()
        if world > 1 and quota is not None and quota < 2 * world:
            flags |= lm.FLAG_BLOCKING_SYNC      # fewer CPUs than busy processes: do not spin while waiting for the GPU
        self.stream_sets = 2 if (world == 1 and not args.no_h2d) else 1
        # --parallelism template-shard (default): every rank holds 1 / N of the bank and sees the same frames, the lists are
        # exchanged; frame-shard: every rank holds the WHOLE bank and matches its own frames, nothing is exchanged
        self.frame_shard = getattr(args, "parallelism", "template-shard") == "frame-shard"
        cfg = lm.default_config(color_only=wl["color_only"], width=W, height=H, device=local_rank, shard_rank=0 if self.frame_shard else rank,
                                shard_size=1 if self.frame_shard else world, frame_slots=max(B, 1) * self.stream_sets, flags=flags)
        self.det = det = lm.Detector(cfg)
        # ---- workload: seeded synthetic frames + fixed-geometry bank
        seed0 = wl["seed_frames"] + (rank * B if self.frame_shard else 0)          # frame-shard: every rank its own frames
        self.frames = [synth.make_frame(W, H, seed=seed0 + i) for i in range(B)]
        q = quantized_from_gpu(det, self.frames[0][0], self.frames[0][1], M)
        self.banks = []
        if wl.get("classes"):
            self.n_total = wl["classes"] * wl["templates_per_class"]
            for c in range(wl["classes"]):
                descs, feats, _ = synth.make_bank(wl["templates_per_class"], M, 2, seed=wl["seed_bank"] + c, size_range=wl["size_range"],
                                                  quantized=q, crop_fraction=0.02, frame_size=(W, H), T0=det.get_T(0))
                self.banks.append(("model%d.ply" % c, descs, feats))
        else:
            descs, feats, _ = synth.make_bank(self.n_total, M, 2, seed=wl["seed_bank"], fixed_l0_size=wl["l0_size"],
                                              quantized=q, crop_fraction=0.1, frame_size=(W, H), T0=det.get_T(0))
            self.banks.append(("synthetic.ply", descs, feats))
        for name, descs, feats in self.banks:
            det.add_class(name, descs, feats)
        self.cls = -1 if len(self.banks) > 1 else 0       # all classes of the bank = upstream's class list of all ids
        if args.no_batch_phases:
            det.set_tuning(lm.TUNE_BATCH_PHASES, 0)
        if args.batch_phases >= 0:
            det.set_tuning(lm.TUNE_BATCH_PHASES, args.batch_phases)
        if args.cblur_variant:
            det.set_tuning(lm.TUNE_CBLUR_VARIANT, args.cblur_variant)
        if args.pyrdown_variant:
            det.set_tuning(lm.TUNE_PYRDOWN_VARIANT, args.pyrdown_variant)
        if args.no_blur_pyr:
            det.set_tuning(lm.TUNE_BLUR_PYR, 0)
        if args.blur_pyr >= 0:
            det.set_tuning(lm.TUNE_BLUR_PYR, args.blur_pyr)
        if args.blur_strip:
            det.set_tuning(lm.TUNE_BLUR_STRIP, args.blur_strip)
        if args.sort_split >= 0:
            det.set_tuning(lm.TUNE_SORT_SPLIT, args.sort_split)
        if args.no_work_weight:
            det.set_tuning(lm.TUNE_WORK_WEIGHT, 0)
        if args.scan_list_order >= 0:
            det.set_tuning(lm.TUNE_SCAN_LIST_ORDER, args.scan_list_order)
        if args.scan_form >= 0:
            det.set_tuning(lm.TUNE_SCAN_FORM, args.scan_form)
        if args.no_prune:
            det.set_scan_variant(8)
            det.set_tuning(lm.TUNE_SCAN_FORM, 1)          # the exhaustive scan is k_scan4's (k_scan1 always prunes): the request-size calibration runs on it
        elif args.scan_variant:
            det.set_scan_variant(args.scan_variant)
        for i, (bgr, depth) in enumerate(self.frames):
            det.upload_frame(i, bgr, depth if M == 2 else None)
        det.upload_wait(-1)
        self.NL = args.lanes
        if B % self.NL or B < self.NL:
            self.NL = 1
        self.Bl = B // self.NL                           # frames per lane and launch
        # a frame's list grows with the bank: 4408 matches at 24 300 templates on one GPU (config 4's whole bank, the weak-scaling denominator)
        self.cap = cap = CAP if self.n_total // max(world, 1) <= 8000 else 4 * CAP
        self.bufs = [(np.zeros((B, cap), lm.MATCH_DTYPE), np.zeros(B, np.int32)) for _ in range(NBUF)]
        self.views = [[(o[l * self.Bl:(l + 1) * self.Bl], cn[l * self.Bl:(l + 1) * self.Bl]) for l in range(self.NL)]
                      for o, cn in self.bufs]
        self.k = 0
        self.last_owned = None
        self.scan_kernel_name = "scan"
        if exchange in ("rccl", "rccl-barrier"):
            port = int(os.environ.get("MASTER_PORT", "29500")) + 1
            # fixed gather capacity: 1024 records per frame and rank on average over a lane-step (the bench bank yields
            # about 530 per frame and 3000-template shard at threshold 80); lists beyond it take the sized second exchange
            det.comm_init(rank, world, os.environ.get("MASTER_ADDR", "127.0.0.1"), port, args.gather_cap or 1024)
        if exchange == "rccl":
            self.gbuf = [(np.zeros(self.Bl * self.cap // 4, lm.MATCH_DTYPE), np.zeros(self.Bl, np.int32)) for _ in range(self.NL)]

    # ------------------------------------------------------------------------------------------
    def run_steps(self, n, after_step=None):
        """n passes of the hot path over the batch, driven by this one host thread: lane l works on the frames of
        slots [l * Bl, (l + 1) * Bl); as soon as a lane's step is collected its next step is enqueued, so the
        streams always have work and their stages overlap.  With the RCCL exchange the lane's all-gathers are part of
        what lm_match_begin_gathered enqueues, and lm_match_end_gathered merges the frames this rank owns."""
        if n <= 0:
            return
        det, thr, NL, Bl = self.det, self.args.threshold, self.NL, self.Bl
        rccl = self.exchange == "rccl"
        begin = det.match_begin_gathered if rccl else det.match_begin
        k0 = self.k
        self.k += n
        for l in range(NL):
            begin(l, l * Bl, Bl, thr, self.cls)
        for k in range(k0, k0 + n):
            for l in range(NL):
                if rccl:
                    o, cn = self.gbuf[l]
                    f0, nf, tot = det.match_end_gathered(l, o, cn)
                    if l == 0:
                        self.last_owned = (f0, nf, int(cn[0]) if nf else 0)
                else:
                    o, cn = self.views[k % NBUF][l]
                    det.match_end(l, self.cap, out=o, counts=cn)
                if k + 1 < k0 + n:
                    begin(l, l * Bl, Bl, thr, self.cls)
            if after_step is not None:
                after_step(k)

    def one_lane_profile(self, steps=10):
        """The same launches (Bl frames each) on ONE lane, nothing running beside them and no exchange: the clean
        per-stage and scan-kernel times (HIP events on the launch stream around every stage)."""
        det = self.det
        det.set_profiling(True)
        for _ in range(steps):
            det.match_begin(0, 0, self.Bl, self.args.threshold, self.cls)
            det.match_end(0, self.cap, out=self.views[0][0][0], counts=self.views[0][0][1])
        prof = det.get_profile()
        det.set_profiling(False)
        # one more launch with the scan's feature counters on: what fraction of the feature loads the pruning keeps
        det.set_scan_stats(True)
        det.match_begin(0, 0, self.Bl, self.args.threshold, self.cls)
        det.match_end(0, self.cap, out=self.views[0][0][0], counts=self.views[0][0][1])
        loaded, total = det.get_scan_stats()
        lane_issued, lane_total = det.get_scan_lane_stats()
        form = det.get_scan_form_stats()
        det.set_scan_stats(False)
        prof["scan1_lanes_per_frame"] = form[3]
        prof["scan1_survivors_per_frame"] = form[2] / self.Bl if form[3] else None
        prof["features_loaded_fraction"] = loaded / total if total else 1.0
        prof["lane_loads_fraction"] = lane_issued / lane_total if lane_total else 1.0
        prof["lane_loads_issued"] = lane_issued
        return prof

    def streaming(self, steps):
        """SURVEY.md 8d's metric as defined: H2D of the frame -> D2H of the sorted matches.  Every step uploads all B
        frames again from pinned host memory (the frame -> slot assignment rotates, so every step matches different
        frames in every slot) into one of two slot sets while the lanes compute on the other (lm_upload_frame_pinned
        on the copy stream, per-slot events).  Returns frames/s and the H2D rate."""
        lm, det, B, NL, Bl, M = self.lm, self.det, self.B, self.NL, self.Bl, self.M
        W, H, thr = self.W, self.H, self.args.threshold
        fb = W * H * 3 + (W * H * 2 if M == 2 else 0)
        pb = lm.PinnedBuffer(B * fb)
        hb = [pb.view(np.uint8, (H, W, 3), offset=i * fb) for i in range(B)]
        hd = [pb.view(np.uint16, (H, W), offset=i * fb + W * H * 3) if M == 2 else None for i in range(B)]
        for i, (bgr, depth) in enumerate(self.frames):
            hb[i][...] = bgr
            if M == 2:
                hd[i][...] = depth

        G = max(1, int(self.args.h2d_group or CONFIGS[self.args.config].get("h2d_group", 1)))

        def upload(s, k):
            # frame (i + 7 k) mod B goes to slot i of the set: runs of G consecutive slots take G consecutive host frames, which
            # lie back to back in the pinned block -- one strided transfer per run (lm_upload_frames_pinned: the larger the DMA,
            # the closer to the link rate), single frames otherwise
            i = 0
            while i < B:
                j = (i + 7 * k) % B
                g = min(G, B - i, B - j)
                if g > 1:
                    det.upload_frames_pinned(s * B + i, g, pb.ptr.value + j * fb, fb)
                else:
                    det.upload_frame_pinned(s * B + i, hb[j], hd[j])
                i += g

        def begin(s):
            for l in range(NL):
                det.match_begin(l, s * B + l * Bl, Bl, thr, self.cls)

        def end(k):
            for l in range(NL):
                o, cn = self.views[k % NBUF][l]
                det.match_end(l, self.cap, out=o, counts=cn)

        counts = []
        for phase, n in (("warm", 3), ("timed", steps)):
            upload(0, 0)
            det.upload_wait(-1)
            det.synchronize()
            t0 = time.perf_counter()
            begin(0)
            for k in range(n):
                if k + 1 < n:
                    upload((k + 1) & 1, k + 1)          # travels while step k computes
                end(k)
                if phase == "timed" and k < 2:
                    counts.append(int(self.bufs[k % NBUF][1].sum()))
                if k + 1 < n:
                    begin((k + 1) & 1)
            det.synchronize()
            dt = time.perf_counter() - t0
        pb.close()
        fps = B * steps / dt
        gbs = fps * fb / 1e9
        return {"value": round(fps, 1), "unit": "detections/s", "steps": steps, "ms_per_step": round(dt / steps * 1e3, 4),
                "h2d_bytes_per_frame": fb, "h2d_GBps": round(gbs, 2), "pcie_peak_GBps": PCIE_PEAK_GBS,
                "frac_of_pcie_peak": round(gbs / PCIE_PEAK_GBS, 4),
                "pcie_bound_detections_per_s": round(PCIE_PEAK_GBS * 1e9 / fb, 1),
                "matches_step0_step1": counts,
                "frames_per_transfer": G,
                "note": "every step uploads all %d frames from pinned host memory (rotating frame -> slot map, runs of up to %d "
                        "consecutive frames per strided transfer) into one of two slot sets while the lanes compute on the other; "
                        "the upload of step 0 is inside the timed region; the link, not the GPU, bounds this mode" % (B, G)}

    def timed_lists(self):
        """Copies of what the LAST step of run_steps left in the lanes' own result buffers -- frames 0..3 of lane 0 and the first frame of every
        other lane -- as {frame index: list}.  cpu_baseline() holds them against the oracle: the acceptance check then looks at the launch forms
        the timed region ran (batch kernels, one scan launch per lane-step), not at a separate single-frame call (VERDICT r5 #1b)."""
        if not self.k or self.exchange == "rccl":
            return {}
        out = {}
        views = self.views[(self.k - 1) % NBUF]
        for l in range(self.NL):
            o, cn = views[l]
            for j in range(min(4, self.Bl) if l == 0 else 1):
                n = int(cn[j])
                out[l * self.Bl + j] = (n, o[j, :min(n, self.cap)].copy())
        return out

    def report(self):
        prof = self.det.get_profile()
        counts = np.array([self.det.last_counts(i) for i in range(self.B)], np.int64).reshape(-1, 2)
        self.list_stats = {"scan_candidates_per_frame": {"mean": round(float(counts[:, 0].mean()), 1), "max": int(counts[:, 0].max())},
                           "refined_matches_per_frame_before_unique": {"mean": round(float(counts[:, 1].mean()), 1), "max": int(counts[:, 1].max())}}
        return {"prof": prof, "scan_load_bytes": self.det.scan_load_bytes(self.cls), "Bl": self.Bl, "NL": self.NL,
                "matches0": int(self.bufs[(self.k - 1) % NBUF][1][0]) if self.k else 0}


def launch_ranks(n, argv):
    """`bench.py --gpus N` started as ONE plain process (no torchrun, WORLD_SIZE unset): this parent -- which imports
    neither the package nor anything that touches HIP -- starts N fresh child processes of itself, one rank per GPU
    (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT in their environment), relays rank 0's one JSON line,
    and fails if any rank fails: a child that dies takes the others with it (a rank that never arrives would leave
    the rest waiting in the rendezvous), and the exit code is non-zero.  Never an os.exec*: the children are new
    processes."""
    port = os.environ.get("MASTER_PORT")
    if not port:
        # a free port for this run whose successor is free as well: port + 1 is the rendezvous of the communicators' ids
        for _ in range(64):
            with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
                sk.bind(("127.0.0.1", 0))
                cand = sk.getsockname()[1]
            if cand >= 65535:
                continue
            try:
                with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk2:
                    sk2.bind(("127.0.0.1", cand + 1))
            except OSError:
                continue
            port = str(cand)
            break
        else:
            sys.stderr.write("bench.py: found no free port pair for the rendezvous\n")
            return 1
    import threading
    procs = []
    captured = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR=os.environ.get("MASTER_ADDR", "127.0.0.1"), MASTER_PORT=port,
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))
    reader = threading.Thread(target=lambda: captured.append(procs[0].stdout.read()), daemon=True)
    reader.start()                                        # rank 0's stdout is drained while it runs (a full pipe would block it)
    failed = None
    pending = set(range(n))
    while pending and failed is None:
        for r in sorted(pending):
            rc = procs[r].poll()
            if rc is None:
                continue
            pending.discard(r)
            if rc != 0:
                failed = (r, rc)
                break
        time.sleep(0.05)
    if failed is not None:
        for r in pending:
            procs[r].kill()
        for pr in procs:
            try:
                pr.wait(timeout=30)
            except Exception:  # noqa: BLE001
                pass
        sys.stderr.write("bench.py: rank %d of %d exited with code %d; the run is void (no JSON line)\n" % (failed[0], n, failed[1]))
        return failed[1] if failed[1] > 0 else 1
    reader.join(timeout=30)
    out = captured[0] if captured else ""
    lines = [l for l in out.splitlines() if l.startswith('{"metric"')]
    if len(lines) != 1:
        sys.stderr.write("bench.py: rank 0 printed %d JSON lines instead of 1\n" % len(lines))
        return 1
    d = json.loads(lines[0])
    cfg = d.get("config", {})
    if d.get("n_gpus") != n or not (cfg.get("rccl_ranks") == n or cfg.get("functional_gloo")):
        sys.stderr.write("bench.py: asked for %d GPUs, the line reports n_gpus=%r rccl_ranks=%r\n"
                         % (n, d.get("n_gpus"), cfg.get("rccl_ranks")))
        return 1
    print(lines[0])
    sys.stdout.flush()
    return 0


CONFIG4_TEMPLATES = 24300    # BASELINE config 4: 162 viewpoints x 15 radii x 10 rotations, sharded over 8 GPUs (3037 / 3038 each)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", type=int, default=2, choices=sorted(CONFIGS),
                    help="2 / 3: SURVEY 8d configs 2 / 3 (the headline is config 2); 5: the hot path of BASELINE configs[4] (8-frame "
                         "batches of 1280x960 RGB-D, 3 classes x 8100 templates)")
    ap.add_argument("--batch", type=int, default=0,
                    help="frames per step (resident in HBM); 0 = the config's default (config 2: 288 = 3 lanes x 96, "
                         "config 3: 256 = 2 lanes x 128: the launch shapes measured fastest, DESIGN.md section 6)")
    ap.add_argument("--lanes", type=int, default=0, choices=(0, 1, 2, 3, 4),
                    help="the step's frames are split over this many of the detector's lanes (HIP streams driven by one "
                         "host thread) so that the stages of one part overlap those of the others; 0 = the config's default")
    ap.add_argument("--templates", type=int, default=0,
                    help="templates per GPU (weak) / in the whole bank (strong); 0 = 3000 (config 2 / 3) on one GPU and, "
                         "for N > 1 with weak scaling, BASELINE config 4's dense viewpoint sphere: 24 300 x N / 8 in the "
                         "whole bank (N = 8: 24 300, 3037 / 3038 per GPU)")
    ap.add_argument("--scaling", default="weak", choices=("weak", "strong"))
    ap.add_argument("--parallelism", default="template-shard", choices=("template-shard", "frame-shard"),
                    help="template-shard (default, SURVEY 8e): the bank is split over the ranks, every rank pre-processes the same frames, the "
                         "per-shard lists are all-gathered and merged; frame-shard: every rank holds the whole bank and matches its OWN frames -- "
                         "no exchange, and the pre-processing (a3-a10, half of a rank's time) scales with N too: the split for batch workloads "
                         "(BASELINE config 5) whose bank fits one GPU")
    ap.add_argument("--threshold", type=float, default=80.0)
    ap.add_argument("--gather-cap", type=int, default=0, help="lm_comm_init recs_per_frame_cap (0 = 1024)")
    ap.add_argument("--byte-responses", action="store_true", help="LM_FLAG_BYTE_RESPONSES: byte scan kernel (A/B)")
    ap.add_argument("--scan-variant", type=int, default=0, help="A/B knob: features per load block of the scan (0: 6, 1: 12, 2: 3)")
    ap.add_argument("--no-prune", action="store_true",
                    help="exhaustive similarity scan: every feature of every template at every position, even where the "
                         "threshold is already out of reach (scan variant bit 3; A/B of the exact pruning)")
    ap.add_argument("--no-batch-phases", action="store_true",
                    help="A/B knob: one launch per pre-processing kernel (eleven per lane-step) instead of the four launches of "
                         "level-fused batch kernels (LM_TUNE_BATCH_PHASES = 0)")
    ap.add_argument("--batch-phases", type=int, default=-1, choices=(-1, 0, 1, 2), help="A/B knob: LM_TUNE_BATCH_PHASES (1: level-fused launches also beside other lanes)")
    ap.add_argument("--cblur-variant", type=int, default=0, help="A/B knob: LM_TUNE_CBLUR_VARIANT (1: one-shot, 3: shared column sums, 4: matrix cores)")
    ap.add_argument("--pyrdown-variant", type=int, default=0, help="A/B knob: LM_TUNE_PYRDOWN_VARIANT (1: k_pyrdown8, 2: row-walking k_pyrdown16)")
    ap.add_argument("--blur-strip", type=int, default=0, choices=(0, 16, 32, 64), help="A/B knob: rows per strip of the level-0 blur (LM_TUNE_BLUR_STRIP)")
    ap.add_argument("--blur-pyr", type=int, default=-1, choices=(-1, 0, 1, 2, 3), help="A/B knob: LM_TUNE_BLUR_PYR (1: blur and pyrDown tiles of a slot back to back, 2: dealt out evenly, 3: by frame size = default)")
    ap.add_argument("--no-blur-pyr", action="store_true", help="A/B knob: level-0 blur and pyrDown as two launches (LM_TUNE_BLUR_PYR = 0)")
    ap.add_argument("--sort-split", type=int, default=-1, choices=(-1, 0, 1, 2), help="A/B knob: LM_TUNE_SORT_SPLIT (0 one workgroup per frame, 1 chunk workgroups + merge launch, 2 adaptive = default)")
    ap.add_argument("--scan-list-order", type=int, default=-1, choices=(-1, 0, 1, 2, 3), help="A/B knob: LM_TUNE_SCAN_LIST_ORDER (0 ascending offsets, 1 round-robin over orientations, 2 descending, 3 farthest-point = default)")
    ap.add_argument("--scan-form", type=int, default=-1, choices=(-1, 0, 1, 2, 3), help="A/B knob: LM_TUNE_SCAN_FORM (0 by cost = default, 1 the nibble scan k_scan4, 2 the bit-plane scan k_scan1)")
    ap.add_argument("--no-work-weight", action="store_true", help="A/B knob: few-frame / batch kernel selection by frame count alone (LM_TUNE_WORK_WEIGHT = 0, r03)")
    ap.add_argument("--no-pose-e2e", action="store_true", help="config 5: skip the pose_e2e leg (PoseDetection::detectBatch end to end, tools/pose_e2e_bench.cpp)")
    ap.add_argument("--pose-e2e-iters", type=int, default=20)
    ap.add_argument("--no-latency", action="store_true", help="skip the `latency` block (one 640x480 colour-only frame per call against 1950 templates: the reference's call pattern)")
    ap.add_argument("--latency-calls", type=int, default=300)
    ap.add_argument("--pose-e2e-threads", type=int, default=0, help="pose_e2e leg: host threads of the facade's pool (0 = one per usable CPU, at most 32)")
    ap.add_argument("--pose-e2e-wrap", default="", help="pose_e2e leg: command prefix for the child process (e.g. a rocprofv3 trace command ending in --)")
    ap.add_argument("--pose-e2e-host-colour", action="store_true", help="pose_e2e leg: also time the colour check on the host (one full-frame mask per match)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-h2d", action="store_true", help="skip the h2d_inclusive (streaming) leg")
    ap.add_argument("--h2d-group", type=int, default=0,
                    help="streaming leg: consecutive frames per H2D transfer (lm_upload_frames_pinned for runs, 1 = one transfer per "
                         "frame; 0 = the config's default: 24 frames = 36.9 MB for config 2, 16 = 59 MB for config 3 -- measured r03: "
                         "1.5 MB transfers reach 49.9 GB/s, 24-frame ones 54.3)")
    ap.add_argument("--functional-gloo", action="store_true",
                    help="functional check of the N > 1 path on a 1-GPU box: every rank uses cuda:0 and the exchange "
                         "goes through torch.distributed/gloo on the host (never a measurement; RCCL refuses two ranks "
                         "on one device)")
    ap.add_argument("--force-rccl", action="store_true",
                    help="use the RCCL exchange even with one rank (single-rank communicator: exercises the whole "
                         "gathered path on a 1-GPU box)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the cpu_baseline sample")
    args = ap.parse_args()

    if CONFIGS[args.config].get("classes") and args.parallelism != "frame-shard":
        args.scaling = "strong"            # config 5's bank is fixed (3 x 8100 templates): N ranks shard it, it does not grow with N
    if not args.batch:
        args.batch = CONFIGS[args.config]["batch"] if not args.lanes else (128 if args.config != 5 else 8) * args.lanes
    if not args.lanes:
        args.lanes = CONFIGS[args.config]["lanes"]
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: become the launcher of N ranks (before anything touches HIP)
        raise SystemExit(launch_ranks(args.gpus, sys.argv[1:]))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("--gpus %d does not match WORLD_SIZE %d: a run must have exactly one rank per GPU asked for"
                         % (args.gpus, world))
    if not 0 <= rank < world:
        raise SystemExit("RANK %d outside WORLD_SIZE %d" % (rank, world))
    if args.parallelism == "frame-shard":
        args.templates_total = args.templates or 3000          # the whole bank on every rank; the frames grow with N (weak scaling)
        args.scaling = "weak"
    elif not args.templates:
        args.templates_total = (CONFIG4_TEMPLATES * world + 4) // 8 if (world > 1 and args.scaling == "weak") else 3000
    else:
        args.templates_total = args.templates * world if args.scaling == "weak" else args.templates
    if args.functional_gloo:
        local_rank = 0
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

    # No torch in this process: torch's wheel bundles its own ROCm 7.0.2 libamdhip64 / libhsa-runtime64 with the same
    # sonames as /opt/rocm's 7.2 ones; once it is imported the matcher runs on THAT runtime, under which the detector's
    # two streams overlap far less (measured r02: 99.9 K detections/s without torch, 88.2 K with it; one lane 92.7 K
    # either way).  The exchange is RCCL called from the C++ library.
    lm = importlib.import_module("line-mod-pipeline_amd")
    # librccl prints a version banner on STDOUT (from its own threads, when the first communicator / collective comes up);
    # stdout carries exactly ONE JSON line, so file descriptor 1 points at stderr for the whole run and the line is written
    # to the saved descriptor at the end
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    frame_shard = args.parallelism == "frame-shard"
    if frame_shard and args.functional_gloo:
        raise SystemExit("--functional-gloo checks the exchange of the template-shard path; frame-shard has none")
    exchange = "gloo" if (args.functional_gloo and world > 1) else ("rccl" if (world > 1 or args.force_rccl) else "none")
    if frame_shard and exchange == "rccl":
        exchange = "rccl-barrier"           # the communicator only carries the barrier and the max of the ranks' times
    runner = Runner(args, rank, world, local_rank, exchange)
    det = runner.det
    rccl_ranks, bus_ids = None, [det.pci_bus_id()]
    if exchange in ("rccl", "rccl-barrier"):
        # the communicator really has one rank per GPU asked for, and the ranks really sit on different GPUs
        crank, cworld = det.comm_info()
        if cworld != args.gpus or crank != rank:
            raise SystemExit("RCCL communicator has rank %d of %d, the run was asked for rank %d of %d GPUs" % (crank, cworld, rank, args.gpus))
        rccl_ranks = cworld
        if world > 32:
            raise SystemExit("more than 32 ranks on one node")
        ids = det.comm_max([float(pci_bus_number(bus_ids[0])) if r == rank else -1.0 for r in range(world)])
        bus_ids = ["%04x:%02x:%02x.%x" % ((int(v) >> 16) & 0xFFFF, (int(v) >> 8) & 0xFF, (int(v) >> 3) & 0x1F, int(v) & 7) for v in ids]
        if len(set(ids)) != world or min(ids) < 0:
            raise SystemExit("the %d ranks do not sit on %d different GPUs: PCI bus ids %s" % (world, world, bus_ids))
    gather = dist = None
    state = {"merged": None}
    if exchange == "gloo":
        import torch.distributed as dist
        distmod = importlib.import_module("line-mod-pipeline_amd.dist")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="gloo", rank=rank, world_size=world)
        gather = distmod.ShardGather(lm.merge_matches, cap=runner.cap, pack_fn=lm.pack_matches, merge_batch_fn=lm.merge_batch)

    def after_step(k):
        o, cn = runner.bufs[k % NBUF]
        state["merged"] = gather.gather_merge_packed(lm.pack_matches(o, cn), cn, owned_only=True)

    def run_steps(n):
        runner.run_steps(n, after_step=after_step if exchange == "gloo" else None)

    def fence():
        # a barrier over the ranks + a device-wide synchronisation on both sides of the timed region
        if exchange in ("rccl", "rccl-barrier"):
            det.comm_barrier()                          # hipDeviceSynchronize + ncclAllReduce + hipDeviceSynchronize
        elif exchange == "gloo":
            det.synchronize()
            dist.barrier()
        else:
            det.synchronize()                           # hipDeviceSynchronize

    run_steps(args.warmup)
    det.set_profiling(True)
    fence()
    t0 = time.perf_counter()
    run_steps(args.steps)
    fence()
    dt = time.perf_counter() - t0
    rep = runner.report()
    timed_lists = runner.timed_lists()                  # (before the legs below reuse the result buffers)
    exch_us, exch_n, exch_fb = det.get_exchange_profile() if exchange == "rccl" else (0.0, 0, 0)
    det.set_profiling(False)
    one_lane = runner.one_lane_profile()
    if exchange in ("rccl", "rccl-barrier"):
        dt = det.comm_max([dt])[0]
    elif exchange == "gloo":
        import torch
        tt = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    prof, Bl, NL = rep["prof"], rep["Bl"], rep["NL"]
    B = args.batch
    fps = B * (world if frame_shard else 1) * args.steps / dt          # frame-shard: every rank answered its own B frames per step
    if exchange == "rccl":
        n_matches0 = runner.last_owned[2] if runner.last_owned else 0        # merged list of the first owned frame
    elif exchange == "gloo":
        merged = state["merged"]
        n_matches0 = len(merged[1][0]) if merged and merged[1] else 0
    else:
        n_matches0 = rep["matches0"]

    # ---- roofline of the dominant kernel (similarity scan)
    lanes1 = one_lane.get("scan1_lanes_per_frame") or 0
    # (lm_get_scan_form_stats: 0 = the nibble scan k_scan4, 1..64 = lanes per frame of the bit-plane scan k_scan1, 1000 + shares per frame = the bit-plane scan
    # with a frame's planes in LDS, k_scanl)
    kernel = "k_scanl" if lanes1 >= 1000 else "k_scan1" if lanes1 else ("k_scan4" if not args.byte_responses else "k_scan")
    runner.scan_kernel_name = kernel
    kept = one_lane.get("features_loaded_fraction", 1.0)
    kept_lanes = one_lane.get("lane_loads_fraction", kept)
    l2_bytes = rep["scan_load_bytes"] * Bl * kept_lanes      # bytes the scan's vector loads really request per launch (16 B per active lane)
    if one_lane.get("lane_loads_issued"):
        l2_bytes = 16.0 * one_lane["lane_loads_issued"]      # the same from the kernel's own count (both scan forms; one launch of Bl frames)
    span_us = prof["stage_us"][1] / max(prof["launches"], 1)
    alg_bytes = prof["scan_bytes"] / max(prof["launches"], 1)
    ol_us = one_lane["stage_us"][1] / max(one_lane["launches"], 1)
    cmeta = counters_meta(args, runner.n_total if frame_shard else runner.n_total // world, Bl)
    ctr, ctr_reason = load_counters(cmeta)
    traffic = traffic_src = None
    if ctr:
        for k, v in ctr["kernels"].items():
            if kernel_is(k, kernel) and "hbm_bytes_per_launch" in v:
                traffic, traffic_src = v["hbm_bytes_per_launch"], ctr["source"]
    l2c = ctr if ctr and any("TCP_TCC_READ_REQ_sum" in v for v in ctr["kernels"].values()) else None

    def rate(nbytes, us):
        return nbytes / (us * 1e-6) / 1e9 if us > 0 else 0.0

    wl_cfg = CONFIGS[args.config]
    first_peak = LDS_B32_PEAK_GBS if kernel == "k_scanl" else L2_PEAK_GBS
    roofline = {
        "bound": "lds" if kernel == "k_scanl" else "l2", "achieved": round(rate(l2_bytes, ol_us), 1), "peak": first_peak, "unit": "GB/s",
        "frac": round(rate(l2_bytes, ol_us) / first_peak, 4), "traffic": traffic, "traffic_source": traffic_src,
        "kernel": kernel, "avg_launch_us": round(ol_us, 2), "frames_per_launch": Bl, "load_bytes_per_launch": l2_bytes,
        "pruning": {"enabled": not args.no_prune and not args.byte_responses, "feature_loads_kept": round(kept, 4),
                    "lane_loads_kept": round(kept_lanes, 4),
                    "note": "exact: a work item stops loading features once partial sum + 4 x features to come cannot "
                            "exceed the raw threshold at any of its positions (same candidate list; bench.py --no-prune "
                            "runs the exhaustive scan); load_bytes_per_launch counts only the loads that were made"},
        "measured": "HIP events on the launch stream around every %s launch of %d launches (%d frames each) on one lane "
                    "with nothing running beside them, in this process right after the timed region; rocprofv3 "
                    "--kernel-trace --stats of `bench.py --lanes 1` agrees (profiles/)" % (kernel, one_lane["launches"], Bl),
        "why_l2": why_l2(runner, wl_cfg, traffic, alg_bytes, ctr, kernel),
        "counter_file": ctr["source"] if ctr else None, "counter_file_reason": ctr_reason,
        "hbm_algorithmic": {
            "bound": "hbm", "achieved": round(rate(alg_bytes, ol_us), 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(rate(alg_bytes, ol_us) / HBM_PEAK_GBS, 4), "algorithmic_bytes_per_launch": alg_bytes,
            "note": "SURVEY.md 8d yard-stick: sum over templates and modalities of features x scanned positions, one "
                    "byte each (ALL features, also those the pruning never loads), divided by the same clean launch "
                    "duration; > 1 because the bytes come from L2 and because of the pruning"},
        "timed_region": {
            "avg_span_us": round(span_us, 2), "lanes": NL,
            "note": "HIP-event span around the scan launch inside the timed, multi-lane run: it contains time in which "
                    "the other lane's kernels hold the chip, so it is not a kernel duration"},
        "stage_us_per_frame_one_lane": dict(zip(["preprocess", "scan", "refine", "sort"],
                                                [round(v / max(one_lane["frames"], 1), 2) for v in one_lane["stage_us"]])),
        "scan1_lanes_per_frame": one_lane.get("scan1_lanes_per_frame"), "scan1_survivors_per_frame": one_lane.get("scan1_survivors_per_frame"),
        "stage_note": "one lane alone on the chip: with LM_TUNE_BATCH_PHASES at its default (auto) the pre-processing of a lone "
                      "lane of 16+ frames runs as level-fused launches, while lanes that run beside others -- the timed region "
                      "with %d lanes -- take one launch per kernel (measured r03: fused wins alone, loses beside other lanes); "
                      "--no-batch-phases forces one launch per kernel everywhere" % NL,
    }
    roofline_refine = None
    if l2c:
        bpr = l2c["bytes_per_request"]

        def from_counters(prefix, us):
            ks = [k for k in l2c["kernels"] if kernel_is(k, prefix)]
            if not ks or us <= 0:
                return None
            c = l2c["kernels"][ks[0]]
            if "TCP_TCC_READ_REQ_sum" not in c:
                return None
            req = c.get("TCP_TCC_READ_REQ_sum", 0.0)
            hit, miss = c.get("TCC_HIT_sum", 0.0), c.get("TCC_MISS_sum", 0.0)
            return {"kernel_in_counter_file": ks[0], "TCP_TCC_READ_REQ_per_launch": req, "bytes_per_request": bpr,
                    "l2_to_l1_bytes_per_launch": req * bpr, "achieved_GBps": round(rate(req * bpr, us), 1),
                    "frac_of_l2_peak": round(rate(req * bpr, us) / L2_PEAK_GBS, 4),
                    "TCP_TOTAL_CACHE_ACCESSES_per_launch": c.get("TCP_TOTAL_CACHE_ACCESSES_sum"),
                    "TCC_hit_rate": round(hit / (hit + miss), 4) if hit + miss else None}

        lc = from_counters(kernel, ol_us)
        if lc and not args.no_prune and not args.scan_variant and kernel != "k_scanl":      # (k_scanl reads its planes from LDS: the L2 -> L1 rate says nothing about it)
            # The line's headline fraction comes from COUNTERS when a committed pass of this very command exists: L1 -> L2 read
            # requests x the calibrated request size (128 B: a request moves a whole line) over this run's clean launch
            # duration.  The r02 model -- bytes the loads NAME, 16 B per active lane -- stays beside it as `load_model`.
            roofline["load_model"] = {"achieved": roofline["achieved"], "frac": roofline["frac"], "unit": "GB/s",
                                      "load_bytes_per_launch": l2_bytes,
                                      "note": "bytes the scan's vector loads request (16 B per lane-load issued, lm_get_scan_lane_stats) over "
                                              "the clean launch duration: the r02 figure; a 128-B line is moved per request, so the L2 -> L1 "
                                              "traffic the counters see is larger"}
            roofline["achieved"], roofline["frac"] = lc["achieved_GBps"], lc["frac_of_l2_peak"]
            lc.update({"source": l2c["source"], "calibration": l2c.get("calibration"),
                       "note": "TCP_TCC_READ_REQ_sum of the committed rocprofv3 --pmc pass of this command (one lane, %d frames per "
                               "launch; tools/collect_counters.sh) x the calibrated request size, over THIS run's clean launch "
                               "duration (HIP events)" % Bl})
            roofline["l2_counters"] = lc
        sq = next((v for k, v in l2c["kernels"].items() if kernel_is(k, kernel) and v.get("SQ_ACTIVE_INST_VALU") and v.get("GRBM_GUI_ACTIVE") and v.get("avg_us")), None)
        if kernel in ("k_scan1", "k_scanl") and sq and not args.no_prune and not args.scan_variant:
            # the bit-plane scan moves a quarter of k_scan4's bytes per position and feature: its binding roof is the vector issue rate.  VALU-active
            # SIMD cycles (SQ_ACTIVE_INST_VALU x 4, counter file of this command) over THIS run's clean scan time -- which contains the survivors'
            # exact sums (k_scan1_exact) -- against 1024 SIMDs x the shader clock under load (GRBM_GUI_ACTIVE of the kernel / 8 XCDs / its duration)
            inst = max(sq.get("GRBM_GUI_ACTIVE_instances", 1), 1)
            clock_ghz = sq["GRBM_GUI_ACTIVE"] / (8.0 if inst == 1 else inst) / sq["avg_us"] / 1e3
            ach = sq["SQ_ACTIVE_INST_VALU"] * 4.0 / (ol_us * 1e3)                  # G SIMD-cycles per second
            if kernel == "k_scanl":
                roofline["lds_rate"] = {"achieved": roofline["achieved"], "peak": LDS_B32_PEAK_GBS, "frac": roofline["frac"], "unit": "GB/s",
                                        "note": "bytes the counting loop reads from LDS (16 per lane and feature: two ds_read2_b32; the fifth dword is a DPP move) over the clean "
                                                "time of the whole scan launch, against the guide's aggregate rate of 4-byte LDS reads (75 TB/s)"}
            else:
                roofline["l2_rate"] = {"achieved": roofline["achieved"], "peak": L2_PEAK_GBS, "frac": roofline["frac"], "unit": "GB/s"}
            roofline.update({"bound": "valu", "achieved": round(ach, 1), "peak": round(N_SIMD * clock_ghz, 1), "unit": "G SIMD-cycles/s",
                             "frac": round(ach / (N_SIMD * clock_ghz), 4),
                             "valu": {"SQ_ACTIVE_INST_VALU_per_launch": sq["SQ_ACTIVE_INST_VALU"], "SQ_INSTS_VALU_per_launch": sq.get("SQ_INSTS_VALU"),
                                      "shader_clock_GHz_under_load": round(clock_ghz, 3), "kernel_avg_us_in_counter_file": sq["avg_us"],
                                      "source": l2c["source"],
                                      "note": ("k_scanl counts misses on one bit per position and orientation from planes held in LDS (lds_rate beside this) and "
                                               "takes the survivors' exact sums from LDS in the same launch; the counting loop is bound by vector issue -- the "
                                               "carry-save adders of the bit-sliced counters and the shift-undo; frac = share of the chip's SIMD cycles in which "
                                               "a vector instruction executes, over the clean time of the whole launch (copies, barriers and second stage included)")
                                      if kernel == "k_scanl" else
                                              "k_scan1 counts misses on one bit per position and orientation: its vector loads ask the L2 for a fraction of "
                                              "what k_scan4's do (l2_rate beside this), and the kernel is bound by vector issue -- the carry-save adders of "
                                              "the bit-sliced counters and the shift-undo; frac = share of the chip's SIMD cycles in which a vector "
                                              "instruction executes, over the clean time of the whole scan stage"}})
        ref_us = one_lane["stage_us"][2] / max(one_lane["launches"], 1)
        rr = from_counters("k_refine<", ref_us)
        if rr:
            roofline_refine = {"bound": "l2", "achieved": rr["achieved_GBps"], "peak": L2_PEAK_GBS, "unit": "GB/s",
                               "frac": rr["frac_of_l2_peak"], "kernel": "k_refine (+ k_refine_plan)", "avg_launch_us": round(ref_us, 2),
                               "frames_per_launch": Bl, "counters": rr, "source": l2c["source"],
                               "note": "similarityLocal + refinement (a14): one wave per candidate, a 16 x 16 patch per feature pulls "
                                       "16-17 lines of 128 B for 256 useful bytes; by the counters neither the L2 (this fraction), nor the "
                                       "L1 tag pipeline, nor the vector ALU is saturated on its own (DESIGN.md section 5)"}
    roofline_pre = preprocess_roofline(ctr, ctr_reason, one_lane, Bl)

    result = None
    h2d = cpu = None
    if world == 1 and not args.no_h2d:
        try:
            h2d = runner.streaming(max(args.steps // 4, 8))
        except Exception as e:                          # the bench line must still be printed
            h2d = {"error": "%s: %s" % (type(e).__name__, e)}
    lat = None
    if rank == 0 and world == 1 and not args.no_latency:
        lat = latency(args, lm)
    pose = None
    if rank == 0 and world == 1 and CONFIGS[args.config].get("classes") and not args.no_pose_e2e:
        pose = pose_e2e(args, runner, lm, 1e6 / fps)
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            cpu = cpu_baseline(args, runner, lm, timed_lists)
        wl = CONFIGS[args.config]
        result = {
            "metric": "detections/sec", "value": round(fps, 1), "unit": "detections/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True, "scaling": args.scaling,
            "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": wl["name"] + ", threshold %g" % args.threshold,
                       "baseline_config": args.config,
                       "templates_per_gpu": runner.n_total if frame_shard else runner.n_total // world, "templates_total": runner.n_total,
                       "frames_per_step": B * (world if frame_shard else 1), "lanes": NL, "frames_per_sec": round(fps, 1), "matches_frame0": n_matches0,
                       "list_lengths": runner.list_stats,
                       "unit_definition": "one detection = one frame answered against the whole %d-template bank "
                                          "(input frames resident in HBM)" % runner.n_total,
                       "template_frames_per_sec": round(fps * runner.n_total, 1),
                       "exchange": {"none": "single GPU", "gloo": "torch.distributed gloo (functional check only)",
                                    "rccl": "2 x ncclAllGather per lane-step from liblinemod_hip.so on the lane's stream; "
                                            "each rank merges the frames it owns",
                                    "rccl-barrier": "none in the data path (frame-shard: whole bank per rank, own frames per rank); the RCCL "
                                                    "communicator carries only the barrier and the max of the ranks' times"}[exchange],
                       "rccl_ranks": rccl_ranks, "pci_bus_ids": bus_ids,
                       "functional_gloo": bool(args.functional_gloo) or None,
                       "exchange_span_us_per_lane_step": round(exch_us / exch_n, 2) if exch_n else None,
                       "exchange_sized_fallbacks": exch_fb if exchange == "rccl" else None,
                       "gather_records_per_frame_cap": (args.gather_cap or 1024) if exchange == "rccl" else None,
                       "exchange_span_note": "HIP events on the lane's stream from behind k_sort_unique to behind the D2H of "
                                             "the gathered lists: k_pack_lists + 2 x ncclAllGather + 2 copies, per lane-step "
                                             "of %d frames (contains the wait for the slowest rank)" % Bl if exch_n else None,
                       "h2d_inclusive": h2d,
                       "parallelism": "%s x%d" % (args.parallelism, world)},
            "roofline": roofline,
            "roofline_refine": roofline_refine,
            "roofline_preprocess": roofline_pre,
            "roofline_pipeline": pipeline_roofline(ctr, ctr_reason, NL, dt / args.steps * 1e3, B, Bl),
            "latency": lat,
            "pose_e2e": pose,
            "counters_meta": cmeta,
            "cpu_baseline": cpu,
        }
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(result) + "\n").encode())
    if exchange == "gloo":
        dist.barrier()
        dist.destroy_process_group()
    det.close()
    return result


HBM_ACHIEVABLE_GBS = 6300.0   # MI355X_MICROARCH.md: measured streaming ceiling of HBM3E on this part (what a copy kernel reaches)
N_SIMD = 1024                 # 256 CUs x 4 SIMDs
PRE_PREFIXES = ("k_blur_pyr", "k_blur_mx", "k_cblur", "k_cgrad", "k_corient", "k_cvote", "k_pyrdown", "k_dnormal", "k_dmedian", "k_lm_", "k_linear_memories",
                "k_phase", "k_bphase", "k_bsplit", "k_pair", "k_color_quantize", "k_depth_quantize", "k_nn_half", "k_pre")


def why_l2(runner, wl, traffic, alg_bytes, ctr, kernel):
    """Why the scan is priced against the L2 rate, from THIS config's working set and -- when a counter file matches -- its
    measured HBM traffic and L2 hit rate (the sentence used to be config 2's for every config)."""
    W1, H1, M = wl["W"] // 2, wl["H"] // 2, runner.M
    lm_bytes = M * 8 * W1 * H1 // 2                              # nibble-packed response memories of the scanned level, per frame
    bank = runner.n_total // max(runner.world, 1) * M * 31 * 4   # u32 offsets of the shard's level-1 features
    msg = ("scanned level per frame: %.2f MB of nibble-packed linear memories (%d modalities x 8 orientations x %d x %d positions / 2); "
           "a wave scans two frames, an XCD works on one slot pair at a time: %.2f MB + %.2f MB of bank offsets against a 4 MB L2 per XCD"
           % (lm_bytes / 1e6, M, W1, H1, 2 * lm_bytes / 1e6, bank / 1e6))
    fits = 2 * lm_bytes + bank <= 4.0e6
    if kernel == "k_scanl":
        planes = M * 8 * W1 * H1 // 8
        msg = ("bit-plane scan with the planes in LDS: a 1024-thread workgroup copies ONE frame's %.1f KB of miss planes (%d modalities x 8 orientations x %d x %d positions "
               "/ 8) into its CU's 160 KB of LDS and counts misses from there, then replaces them by the frame's %.1f KB of spread bytes and takes the survivors' exact sums "
               "from LDS as well -- the loop reads nothing from L2 but the templates' feature lists; the binding roof is vector issue (valu), the LDS read rate is reported "
               "as lds_rate" % (planes / 1e3, M, W1, H1, M * W1 * H1 / 1e3))
        fits = True
    if kernel == "k_scan1":
        planes = M * 8 * W1 * H1 // 8
        msg = ("bit-plane scan: the scanned level per frame is %.2f MB of miss planes (%d modalities x 8 orientations x %d x %d positions / 8) next to the "
               "%.2f MB of nibble memories, which only the survivors' exact sums (k_scan1_exact) read; a wave scans a group of frames, an XCD works on "
               "one group at a time; the L2 -> L1 figure is reported as l2_rate, the binding roof is vector issue (valu)" % (planes / 1e6, M, W1, H1, lm_bytes / 1e6))
        fits = True
    if traffic is not None and alg_bytes:
        msg += "; measured HBM-side traffic of the kernel %.1f MB per launch = %.2f %% of the algorithmic bytes" % (traffic / 1e6, 100.0 * traffic / alg_bytes)
    hit = None
    if ctr:
        for k, v in ctr["kernels"].items():
            if kernel_is(k, kernel) and v.get("TCC_HIT_sum") is not None and v.get("TCC_MISS_sum") is not None and v["TCC_HIT_sum"] + v["TCC_MISS_sum"] > 0:
                hit = v["TCC_HIT_sum"] / (v["TCC_HIT_sum"] + v["TCC_MISS_sum"])
                msg += ", measured L2 hit rate %.1f %%" % (100.0 * hit)
    if kernel in ("k_scan1", "k_scanl"):
        pass
    elif fits:
        msg += ("; the working set fits the L2, so the kernel is bound by what its vector loads request from the L2s against the guide's 34.5 TB/s "
                "aggregate L2 rate")
    elif hit is not None and hit >= 0.95:
        msg += ("; the nominal working set exceeds one XCD's L2, yet the requests hit it (work items are ordered by template and chunk, so "
                "neighbouring waves touch the same lines of the same frame pair): the L2 -> L1 line rate is the binding roof here too")
    else:
        msg += ("; the working set does NOT fit one XCD's L2 and no counter file of this command says how often the requests hit it: the "
                "L2-rate fraction is an upper-bound yard-stick for this config -- `traffic` against the 8 TB/s HBM peak is reported beside it")
    return msg


def is_scan_kernel(k):
    """the similarity scan's main kernel in a counter file (k_scan, k_scan4<..>, k_scan1 -- not k_scan1_exact, its short second half)"""
    return k.startswith("k_scan") and not k.startswith("k_scan1_exact")


def kernel_is(k, name):
    """counter-file key k names kernel `name` (template arguments aside); a name that ends in '<' is a prefix"""
    return k == name or k.startswith(name + "<") or (name.endswith("<") and k.startswith(name))


def preprocess_roofline(ctr, reason, one_lane, Bl):
    """Per-kernel roofline entries of the pre-processing (a3-a10) and of every other kernel that takes >= 5 % of a one-lane step,
    from the committed counter file of THIS command (kernel stats + PMC passes): HBM-side bytes / clean duration against the
    8 TB/s peak and the 6.3 TB/s achievable rate, vector-ALU activity against the SIMDs' cycles; the bound is named from the
    numbers."""
    if not ctr:
        return {"kernels": None, "reason": reason}
    ks = ctr["kernels"]
    step_us = sum(v.get("avg_us", 0.0) * v.get("calls", 0) for v in ks.values() if v.get("calls", 0) >= 8)
    calls = [v["calls"] for k, v in ks.items() if is_scan_kernel(k) and "calls" in v]
    steps = max(calls) if calls else 1
    # shader clock under load: GRBM_GUI_ACTIVE of the longest kernel (the scan; the counter comes as ONE row per dispatch that sums the 8
    # XCDs) over its duration -- short kernels' GUI_ACTIVE contains launch overhead, so every kernel's cycles are its duration x this clock
    clock_ghz = None
    for k, v in ks.items():
        if is_scan_kernel(k) and v.get("GRBM_GUI_ACTIVE") and v.get("avg_us"):
            inst = max(v.get("GRBM_GUI_ACTIVE_instances", 1), 1)
            clock_ghz = v["GRBM_GUI_ACTIVE"] / (8.0 if inst == 1 else inst) / v["avg_us"] / 1e3
    out, pre_us, pre_bytes = [], 0.0, 0.0
    for k, v in sorted(ks.items(), key=lambda kv: -kv[1].get("avg_us", 0) * kv[1].get("calls", 0)):
        if "avg_us" not in v or v.get("calls", 0) < 8 or k.startswith("__amd"):
            continue
        per_step = v["calls"] / steps
        is_pre = k.startswith(PRE_PREFIXES)
        if is_pre:
            pre_us += v["avg_us"] * per_step
            pre_bytes += v.get("hbm_bytes_per_launch", 0.0) * per_step
        if v["avg_us"] * v["calls"] < 0.05 * step_us and not is_pre:
            continue
        e = {"kernel": k, "stage": "preprocess" if is_pre else "match", "avg_us": v["avg_us"], "launches_per_step": round(per_step, 2),
             "share_of_one_lane_step": round(v["avg_us"] * v["calls"] / step_us, 4) if step_us else None}
        us = v["avg_us"]
        if "hbm_bytes_per_launch" in v and us > 0:
            gbs = v["hbm_bytes_per_launch"] / (us * 1e-6) / 1e9
            e.update({"hbm_bytes_per_launch": v["hbm_bytes_per_launch"], "hbm_read_bytes_per_launch": v.get("hbm_read_bytes_per_launch"),
                      "hbm_write_bytes_per_launch": v.get("hbm_write_bytes_per_launch"), "hbm_GBps": round(gbs, 1),
                      "frac_of_hbm_peak": round(gbs / HBM_PEAK_GBS, 4), "frac_of_hbm_achievable": round(gbs / HBM_ACHIEVABLE_GBS, 4)})
        if "TCP_TCC_READ_REQ_sum" in v and us > 0:
            l2 = v["TCP_TCC_READ_REQ_sum"] * ctr.get("bytes_per_request", 128) / (us * 1e-6) / 1e9
            e.update({"l2_read_GBps": round(l2, 1), "frac_of_l2_peak": round(l2 / L2_PEAK_GBS, 4)})
        if "SQ_ACTIVE_INST_VALU" in v and clock_ghz:
            cyc = us * clock_ghz * 1e3                                   # shader cycles of the clean launch
            e.update({"valu_busy": round(v["SQ_ACTIVE_INST_VALU"] * 4.0 / N_SIMD / cyc, 4),
                      "valu_insts_per_launch": v.get("SQ_INSTS_VALU"),
                      "valu_insts_per_simd_cycle": round(v.get("SQ_INSTS_VALU", 0.0) / N_SIMD / cyc, 4),
                      "kernel_cycles": round(cyc)})
        fr_h, fr_v, fr_l = e.get("frac_of_hbm_achievable", 0.0), e.get("valu_busy", 0.0), e.get("frac_of_l2_peak", 0.0)
        best = max((fr_h, "hbm"), (fr_v, "valu"), (fr_l, "l2"))
        e["bound"] = best[1] if best[0] >= 0.6 else "none saturated (latency / occupancy): hbm %.2f, valu %.2f, l2 %.2f" % (fr_h, fr_v, fr_l)
        out.append(e)
    live = one_lane["stage_us"][0] / max(one_lane["launches"], 1)
    return {"kernels": out, "source": ctr["source"],
            "preprocess_sum_of_kernels_us_per_launch": round(pre_us, 1), "preprocess_live_us_per_launch": round(live, 1),
            "preprocess_hbm_bytes_per_frame": round(pre_bytes / Bl), "frames_per_launch": Bl,
            "shader_clock_GHz_under_load": round(clock_ghz, 3) if clock_ghz else None,
            "definitions": "per kernel from the committed counter file of this very command (one lane, one launch per kernel; meta must equal "
                           "this run's, counters_meta): hbm_* = (2 x FETCH_SIZE + WRITE_SIZE) KB per launch / clean avg duration, against the 8 TB/s "
                           "peak and the 6.3 TB/s a streaming kernel achieves; valu_busy = SQ_ACTIVE_INST_VALU x 4 / (1024 SIMDs x kernel "
                           "cycles), kernel cycles = clean duration x the shader clock under load (GRBM_GUI_ACTIVE of the scan / 8 XCDs / its "
                           "duration); valu_insts_per_simd_cycle = SQ_INSTS_VALU / (1024 x cycles) "
                           "(the issue classes of DESIGN.md section 5 cost 1 / 2.7 and 1 / 4.5 per cycle); l2_* = TCP_TCC_READ_REQ x 128 B; "
                           "bound = the largest of (hbm achievable, valu busy, l2) if >= 0.6; preprocess_live = HIP events of this run"}


def pci_bus_number(s):
    """'0000:c1:00.0' -> domain << 16 | bus << 8 | device << 3 | function (exact in a double)."""
    dom, bus, rest = s.strip().split(":")
    dev, fn = rest.split(".")
    return (int(dom, 16) << 16) | (int(bus, 16) << 8) | (int(dev, 16) << 3) | int(fn, 16)


def pose_e2e(args, runner, lm, hot_path_us_per_frame_resident):
    """BASELINE configs[4] as it is worded -- "end-to-end PoseDetection incl. depth/color check" (VERDICT r3 #9): the C++ facade
    lmamd::PoseDetection::detectBatch (tools/pose_e2e_bench.cpp, built here with g++ against liblinemod_hip.so and run as a child
    process) on this run's own bank and its first 8 frames -- principal-point shift on the host, upload, one class-list match on
    the GPU, then the reference's post-processing of every (class, frame).  The bank travels as the library's compact bank file;
    the templates' poses (the reference's raw 48-byte Template records, HighLevelLinemod.h:130-148) are synthetic: bounding box =
    the template's level-0 size, median depth and camera distance spread over the frames' depth range."""
    import hashlib
    import struct
    import tempfile
    try:
        tools = os.path.join(ROOT, "tools", "pose_e2e_bench.cpp")
        host = sorted(os.path.join(ROOT, "line-mod-pipeline_amd", "host", f) for f in os.listdir(os.path.join(ROOT, "line-mod-pipeline_amd", "host")) if f.endswith(".cpp"))
        h = hashlib.sha256()
        for path in [tools] + host + [lm.LIB_PATH]:
            h.update(open(path, "rb").read())
        exe = os.path.join(tempfile.gettempdir(), "lm_pose_e2e_bench_" + h.hexdigest()[:16])
        if not os.path.exists(exe):
            subprocess.check_call(["g++", "-std=c++17", "-O2", "-pthread", "-o", exe + ".tmp", tools] + host +
                                  ["-L" + os.path.dirname(lm.LIB_PATH), "-llinemod_hip", "-Wl,-rpath," + os.path.dirname(lm.LIB_PATH)])
            os.replace(exe + ".tmp", exe)
        nf = min(8, len(runner.frames))
        with tempfile.TemporaryDirectory() as td:
            bank, pose, raw = os.path.join(td, "bench.bank"), os.path.join(td, "poses.bin"), os.path.join(td, "frames.raw")
            runner.det.save_bank(bank)
            rng = np.random.default_rng(99)
            with open(pose, "wb") as f:
                f.write(struct.pack("<I", len(runner.banks)))
                for _, descs, _ in runner.banks:
                    per = 2 * runner.M
                    d0 = descs[0::per]                                   # level 0, modality 0 of every template
                    rec = np.zeros(len(d0), np.dtype([("t", "<f4", 3), ("q", "<f4", 4), ("bb", "<i4", 4), ("md", "<u2"), ("pad", "<u2")]))
                    rec["t"][:, 2] = -rng.uniform(600, 800, len(d0)); rec["q"][:, 3] = 1.0
                    rec["bb"][:, 2] = d0["width"]; rec["bb"][:, 3] = d0["height"]
                    rec["md"] = rng.integers(500, 1200, len(d0))
                    assert rec.dtype.itemsize == 48
                    f.write(struct.pack("<Q", len(rec))); f.write(rec.tobytes())
            with open(raw, "wb") as f:
                for bgr, depth in runner.frames[:nf]:
                    f.write(np.ascontiguousarray(bgr).tobytes()); f.write(np.ascontiguousarray(depth, np.uint16).tobytes())
            keep = os.environ.get("LM_POSE_E2E_KEEP")          # experiments: keep the child's inputs and binary for runs by hand
            if keep:
                import shutil
                os.makedirs(keep, exist_ok=True)
                for src in (bank, pose, raw, exe):
                    shutil.copy(src, os.path.join(keep, os.path.basename(src) if src != exe else "pose_e2e_bench"))
            out = {}
            for mode in (0, 1) if args.pose_e2e_host_colour else (0,):
                r = subprocess.run(args.pose_e2e_wrap.split() + [exe, bank, pose, raw, str(runner.W), str(runner.H), str(nf), str(args.threshold), str(args.pose_e2e_iters), str(mode), str(args.pose_e2e_threads)],
                                   capture_output=True, text=True, timeout=900)
                if r.returncode != 0:
                    return {"error": "pose_e2e_bench exited with %d: %s" % (r.returncode, r.stderr[-500:])}
                out["host" if mode else "gpu"] = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
        g = out["gpu"]
        ser, pip, pin = g["serial"], g["pipelined"], g["pipelined_pinned"]
        total = pip["us_per_frame"]
        res = {"value": round(1e6 / total, 1),
               "unit": "frames/s through PoseDetection::detectBatchBegin / detectBatchEnd, pageable frames, steady state (batches of %d frames, up to three in flight; %d host threads)" % (nf, g["host_threads"]),
               "us_per_frame": total,
               "us_per_frame_serial": ser["us_per_frame"], "us_per_frame_pipelined_pinned_frames": pin["us_per_frame"],
               "poses_identical_across_passes": g["poses_identical_across_passes"],
               "share_of_wall_pipelined": {"gpu_hot_path": pip["gpu_share_of_wall"], "pcie_link": pip["link_share_of_wall"],
                                           "host_in_begin_staging_and_enqueue": round(pip["in_begin_us_per_frame"] / total, 4),
                                           "host_waiting_for_the_gpu": round(pip["waiting_for_the_gpu_us_per_frame"] / total, 4),
                                           "host_post_processing": round(pip["post_us_per_frame"] / total, 4)},
               "serial": ser, "pipelined": pip, "pipelined_pinned": pin,
               "hot_path_us_per_frame_in_the_timed_region_above": round(hot_path_us_per_frame_resident, 2),
               "matches_per_frame": g["matches_per_frame"], "grouped_poses_per_frame": g["grouped_poses_per_frame"],
               "classes": g["classes"], "templates": g["templates"], "frames": nf, "iterations": g["iterations"], "host_threads": g["host_threads"],
               "note": "the reference's call pattern (PoseDetection.cpp:45-126, HighLevelLinemod.cpp:157-175,206-253,424-515) on the bench's bank and frames, three passes over the same "
                       "batches with bit-identical poses: serial = one detectBatch at a time (r04's figure: shift + upload + match + post-processing as a sum); pipelined = "
                       "detectBatchBegin(k + 2) before detectBatchEnd(k) on three slot sets / lanes: the staging copies (pool, shift applied while copying), the transfer and the "
                       "GPU hot path of batch k + 1 run behind the host post-processing of batch k; pipelined_pinned = the same with frames in pinned memory (row-offset DMA copy, no "
                       "staging).  Post-processing = per-frame grouping on the pool, the colour counts of the whole batch in one GPU call on the colour-check stream, then depth "
                       "check + poses of the independent match groups on the pool (by-part figures are CPU time summed over the threads).  gpu_hot_path = HIP-event spans of "
                       "a3-a15 per lane-step; pcie_link = the batch's bytes over the link alone, measured in the same process.  Synthetic template poses whose median depths "
                       "rarely pass the depth check, so nearly every match of every surviving group is tested -- a pessimistic load for the reference's nth_element over the "
                       "template's bounding box per tested match; `value` of the line is the hot path alone with resident frames and three lanes"}
        if "host" in out:
            res["with_host_colour_check_us_per_frame_pipelined"] = out["host"]["pipelined"]["us_per_frame"]
        return res
    except Exception as e:  # the bench line must still be printed
        return {"error": "%s: %s" % (type(e).__name__, e)}


def latency(args, lm):
    """The reference's own call pattern (VERDICT r4 #5b): ONE 640 x 480 frame per detect() (detector.cpp:17-45), the shipped
    colour-only modality and bank size (linemod_settings.yml:20-27: 13 viewpoints x 15 radii x 10 in-plane rotations = 1 950
    templates), one lm_match per frame -- resident frame (lm_match_slot), frame in pinned host memory, frame in pageable memory.
    Median and 95th percentile of `--latency-calls` calls each, a different frame every call."""
    import time
    W, H, NT = 640, 480, 1950
    try:
        synth = importlib.import_module("line-mod-pipeline_amd.synth")
        d = lm.Detector(lm.default_config(color_only=True, width=W, height=H, frame_slots=8))
        if args.scan_form >= 0:
            d.set_tuning(lm.TUNE_SCAN_FORM, args.scan_form)
        frames = [synth.make_frame(W, H, seed=1234 + i) for i in range(8)]
        q = quantized_from_gpu(d, frames[0][0], None, 1)
        descs, feats, _ = synth.make_bank(NT, 1, 2, seed=4321, quantized=q, crop_fraction=0.1, frame_size=(W, H), T0=d.get_T(0))
        d.add_class("lagergehaeuse.ply", descs, feats)
        out = np.zeros(1 << 16, lm.MATCH_DTYPE)
        for i in range(8):
            d.upload_frame(i, frames[i][0], None)
        pb = lm.PinnedBuffer(8 * W * H * 3)
        pf = [pb.view(np.uint8, (H, W, 3), offset=k * W * H * 3) for k in range(8)]
        for k in range(8):
            pf[k][...] = frames[k][0]
        n = max(int(args.latency_calls), 20)

        def timed(fn):
            for k in range(20):
                fn(k)
            ts = np.empty(n)
            for k in range(n):
                t = time.perf_counter()
                fn(k)
                ts[k] = time.perf_counter() - t
            return {"median_us": round(float(np.median(ts)) * 1e6, 1), "p95_us": round(float(np.percentile(ts, 95)) * 1e6, 1), "min_us": round(float(ts.min()) * 1e6, 1)}
        lists = [len(d.match_slot(k, args.threshold, 0, out=out)) for k in range(8)]
        res = {"resident_frame": timed(lambda k: d.match_slot(k % 8, args.threshold, 0, out=out)),
               "pinned_host_frame": timed(lambda k: d.match(pf[k % 8], None, args.threshold, 0, out=out)),
               "pageable_host_frame": timed(lambda k: d.match(frames[k % 8][0], None, args.threshold, 0, out=out))}
        # where a resident frame's time goes: HIP-event spans of the four stages (same call, profiling on)
        d.set_profiling(True)
        for k in range(50):
            d.match_slot(k % 8, args.threshold, 0, out=out)
        prof = d.get_profile()
        d.set_profiling(False)
        pb.close([d])
        d.close()
        res.update({"calls": n, "workload": "one 640x480 frame per call, ColorGradient only (T = {2, 8}), %d templates of variable geometry, threshold %g" % (NT, args.threshold),
                    "matches_per_frame": lists,
                    "gpu_stage_us_resident": dict(zip(["preprocess", "scan", "refine", "sort"], [round(v / max(prof["launches"], 1), 1) for v in prof["stage_us"]])),
                    "note": "wall time of ONE synchronous call on one host thread (time.perf_counter around lm_match_slot / lm_match through ctypes, caller-owned result "
                            "buffer): the reference's loop (detector.cpp:17-45 -> PoseDetection::detect -> HighLevelLineMOD::detectTemplate -> Detector::match) with the "
                            "shipped settings (linemod_settings.yml:20-27).  Few frames take the latency launch shape: the pre-processing as one launch per dependency "
                            "level (k_phase), copies inline on the compute stream.  gpu_stage_us_resident = HIP-event spans inside the resident-frame call; the rest of "
                            "its wall time is launch + synchronisation overhead of the host (about 12 launches per call)"})
        return res
    except Exception as e:  # the bench line must still be printed
        return {"error": "%s: %s" % (type(e).__name__, e)}


def pipeline_roofline(ctr, reason, NL, ms_per_step, frames_per_step, Bl):
    """The whole step against the vector-issue roof (VERDICT r4 #5a): the hot kernels sit within 30 % of the L2 AND the VALU roofs, and
    what bounds the three-lane step is the vector ALU -- VALU-active cycles of every kernel of a lane-step (SQ_ACTIVE_INST_VALU x 4 from
    the committed counter file of this command) x lanes, over the SIMD cycles the step lasts."""
    if not ctr:
        return {"frac": None, "reason": reason}
    ks = ctr["kernels"]
    clock_ghz = None
    for k, v in ks.items():
        if is_scan_kernel(k) and v.get("GRBM_GUI_ACTIVE") and v.get("avg_us"):
            inst = max(v.get("GRBM_GUI_ACTIVE_instances", 1), 1)
            clock_ghz = v["GRBM_GUI_ACTIVE"] / (8.0 if inst == 1 else inst) / v["avg_us"] / 1e3
    calls = [v["calls"] for k, v in ks.items() if is_scan_kernel(k) and "calls" in v]
    steps = max(calls) if calls else 1
    if not clock_ghz:
        return {"frac": None, "reason": "no shader clock in the counter file"}
    active = insts = 0.0
    for k, v in ks.items():
        if k.startswith("__amd") or "SQ_ACTIVE_INST_VALU" not in v or v.get("calls", 0) < 8:
            continue
        per_step = v["calls"] / steps
        active += v["SQ_ACTIVE_INST_VALU"] * 4.0 * per_step
        insts += v.get("SQ_INSTS_VALU", 0.0) * per_step
    lane_steps = frames_per_step / float(Bl)
    cycles = ms_per_step * 1e-3 * clock_ghz * 1e9 * N_SIMD
    return {"bound": "valu", "frac": round(active * lane_steps / cycles, 4), "valu_wave_insts_per_lane_step": round(insts),
            "valu_active_simd_cycles_per_lane_step": round(active), "cycles_per_valu_inst": round(active / insts, 2) if insts else None,
            "lane_steps_per_step": lane_steps, "shader_clock_GHz_under_load": round(clock_ghz, 3), "simds": N_SIMD, "source": ctr["source"],
            "note": "sum over the kernels of one lane-step (counter file of this command, one lane, %d frames per launch) of SQ_ACTIVE_INST_VALU x 4 = SIMD cycles "
                    "in which a vector instruction executes, x the lane-steps of a step, over (1024 SIMDs x shader clock x ms_per_step): how much of the chip's vector "
                    "issue time the timed step uses.  Perfect overlap of the lanes would reach 1.0; what is left is launch boundaries, tails and the kernels "
                    "that wait for memory" % Bl}


def kernel_source_sha16():
    """Identity of the kernel sources a counter file was collected with: sha256 over the product's csrc files, first 16 hex digits."""
    import glob
    import hashlib
    h = hashlib.sha256()
    for path in sorted(glob.glob(os.path.join(ROOT, "line-mod-pipeline_amd", "csrc", "*"))):
        if path.endswith((".hip", ".h", ".cpp")):
            h.update(os.path.basename(path).encode() + b"\0" + open(path, "rb").read())
    return h.hexdigest()[:16]


COUNTER_KEYS = ("baseline_config", "frames_per_launch", "threshold", "templates_per_gpu", "scan_variant", "scan_form", "no_prune", "byte_responses",
                "kernel_source_sha16")


def counters_meta(args, templates_per_gpu, frames_per_launch):
    """What a committed counter file must agree with before bench.py lets it describe this run."""
    return {"baseline_config": args.config, "frames_per_launch": frames_per_launch, "threshold": args.threshold,
            "templates_per_gpu": templates_per_gpu, "scan_variant": args.scan_variant, "scan_form": max(args.scan_form, 0), "no_prune": bool(args.no_prune),
            "byte_responses": bool(args.byte_responses), "kernel_source_sha16": kernel_source_sha16()}


def load_counters(meta, profiles_dir=None):
    """The newest profiles/*_counters_c<config>.json (tools/collect_counters.sh, profiles/summarize_counters.py) whose `meta`
    equals this run's in every key of COUNTER_KEYS -- same workload, same launch shape, same scan variant, same kernel sources.
    Returns (counters or None, reason): PMC counters cannot be read from inside the process, so without a matching file the line
    carries no counter-based figure and says why."""
    import glob
    paths = sorted(glob.glob(os.path.join(profiles_dir or os.path.join(ROOT, "profiles"), "*_counters_c%d.json" % meta["baseline_config"])), reverse=True)
    if not paths:
        return None, "no profiles/*_counters_c%d.json committed" % meta["baseline_config"]
    why = []
    for path in paths:
        try:
            d = json.load(open(path))
        except Exception as e:  # noqa: BLE001
            why.append("%s: unreadable (%s)" % (os.path.basename(path), e))
            continue
        m = d.get("meta") or {}
        diff = [k for k in COUNTER_KEYS if m.get(k) != meta[k]]
        if not diff and d.get("kernels"):
            d["source"] = os.path.relpath(path, ROOT)
            return d, None
        why.append("%s differs in %s" % (os.path.basename(path), ", ".join("%s (file %r, run %r)" % (k, m.get(k), meta[k]) for k in diff)))
    return None, "; ".join(why)


def cgroup_cpus():
    """CPUs' worth of time the container may use (cgroup v2 cpu.max), None when unlimited / unknown."""
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        return None if q == "max" else max(1, int(round(int(q) / int(p))))
    except Exception:  # noqa: BLE001
        return None


def cpu_baseline(args, runner, lm, timed_lists=None):
    """The CPU oracle (kind "port": a restatement, the reference itself cannot be built here) on this host's cores,
    bounded sample, same frames and bank.  Before the timing is accepted the oracle's lists must equal (i) what the TIMED lanes wrote
    into their own result buffers in the last timed step (frames 0..3 of lane 0, the first frame of every other lane: Runner.timed_lists)
    and (ii) the lists of four single-frame lm_match_slot calls (the latency launch shape)."""
    try:
        from oracle import oracle as O
        lib = O.build(arch="-march=native")
        cores = os.cpu_count() or 1
        color_only = runner.M == 1
        orc = O.Detector(color_only=color_only, lib_path=lib)
        for name, descs, feats in runner.banks:
            orc.add_class(name, descs, feats)
        cls = runner.cls
        det, frames, thr = runner.det, runner.frames, args.threshold
        expected = {}

        def oracle_list(i):
            if i not in expected:
                bgr, depth = frames[i]
                expected[i] = orc.match(bgr, None if color_only else depth, thr, cls, threads=min(cores, 16), cap=1 << 18)
            return expected[i]
        timed_lists = timed_lists or {}
        for i, (n, got) in sorted(timed_lists.items()):
            exp = oracle_list(i)
            if n != len(exp) or got.tobytes() != exp.tobytes():
                return {"error": "the timed region's own list of frame %d (lane %d, %d records) differs from the oracle's (%d records): "
                                 "timing not accepted" % (i, i // runner.Bl, n, len(exp))}
        for i in range(min(4, len(frames))):
            bgr, depth = frames[i]
            det.upload_frame(i, bgr, None if color_only else depth)
            gpu = det.match_slot(i, thr, cls)
            exp = oracle_list(i)
            if gpu.tobytes() != exp.tobytes():
                return {"error": "GPU match list of frame %d differs from the oracle: timing not accepted" % i}
        checked = ("the TIMED lanes' own result buffers of the last timed step (frames %s: %d-frame launches, %s) and four single-frame "
                   "lm_match_slot calls, all identical to the oracle's lists" % (
                       ", ".join(str(i) for i in sorted(timed_lists)), runner.Bl, runner.scan_kernel_name)
                   ) if timed_lists else "four single-frame lm_match_slot calls identical to the oracle's lists (no timed buffers: exchange path)"
        bgr, depth = frames[0]
        depth = None if color_only else depth
        # all logical CPUs is not always fastest (SMT / cgroup CPU quota): take the best of a few team sizes
        cand = {cores, max(cores // 2, 1), max(cores // 4, 1)}
        quota = cgroup_cpus()
        if quota:
            cand |= {min(cores, quota), min(cores, 2 * quota)}
        best_t, threads = None, cores
        for th in sorted(cand):
            t1 = time.perf_counter()
            orc.match(bgr, depth, thr, cls, threads=th)
            t = time.perf_counter() - t1
            if best_t is None or t < best_t:
                best_t, threads = t, th
        t0 = time.perf_counter()
        n = 0
        while True:
            b, d = frames[n % len(frames)]
            orc.match(b, None if color_only else d, thr, cls, threads=threads)
            n += 1
            if time.perf_counter() - t0 >= args.cpu_seconds or n >= 5000:
                break
        dt = time.perf_counter() - t0
        t1 = time.perf_counter()
        orc.match(bgr, depth, thr, cls, threads=1)
        single = time.perf_counter() - t1
        # the same frames through the scalar loop shape (one bounds check per byte of the similarity sums): what r01 / r02
        # reported; the default above hoists the check so that the byte adds vectorise like upstream's SSE path
        O.set_scan_mode(0, lib)
        try:
            t1 = time.perf_counter()
            k = 0
            while time.perf_counter() - t1 < max(args.cpu_seconds / 4, 2.0) and k < 1000:
                b, d = frames[k % len(frames)]
                ms = orc.match(b, None if color_only else d, thr, cls, threads=threads)
                k += 1
            scalar_rate = k / (time.perf_counter() - t1)
            t1 = time.perf_counter()
            orc.match(bgr, depth, thr, cls, threads=1)
            scalar_single = time.perf_counter() - t1
        finally:
            O.set_scan_mode(1, lib)
        return {"value": round(n / dt, 3), "unit": "detections/s", "cores": threads, "kind": "port",
                "scalar_loops": {"value": round(scalar_rate, 3), "unit": "detections/s", "cores": threads,
                                 "single_thread_s_per_frame": round(scalar_single, 4),
                                 "note": "same oracle with one bounds check per byte of the similarity sums (the r01 / r02 figure)"},
                "single_thread_s_per_frame": round(single, 4),
                "sample": "%d full frames (a3-a15, same bank of %d templates) in %.1f s; OpenMP over templates and "
                          "over image rows, %d threads (fastest of {1/4, 1/2, all} of %d logical CPUs and {1, 2} x the "
                          "cgroup CPU quota of %s), byte adds of the similarity sums vectorised by the compiler "
                          "(-O3 -march=native, bounds check hoisted); upstream-faithful single-thread run: %.3f s/frame; accepted after "
                          "comparing %s" % (n, runner.n_total, dt, threads, cores, quota or "none", single, checked),
                "timed_buffers_checked": sorted(timed_lists)}
    except Exception as e:  # the bench line must still be printed
        return {"error": "%s: %s" % (type(e).__name__, e)}


if __name__ == "__main__":
    main()
