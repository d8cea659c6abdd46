"""r06 probe: k_scanl's launch time by shares per frame (LM_SCANL_R) and batch size, config 2's workload."""
import importlib, sys, os, subprocess
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    lm = importlib.import_module("line-mod-pipeline_amd")
    synth = importlib.import_module("line-mod-pipeline_amd.synth")
    sizes = [int(a) for a in sys.argv[2:]]
    W, H, M = 640, 480, 2
    NB = max(sizes)
    d = lm.Detector(lm.default_config(color_only=False, width=W, height=H, frame_slots=NB))
    frames = [synth.make_frame(W, H, seed=1234 + i) for i in range(NB)]
    d.upload_frame(0, frames[0][0], frames[0][1]); d.prepare_slot(0)
    q = {(l, m): d.debug_read(0, 0, l, m).reshape(H >> l, W >> l) for l in range(2) for m in range(M)}
    descs, feats, _ = synth.make_bank(3000, M, 2, seed=4321, fixed_l0_size=(96, 96), quantized=q, crop_fraction=0.1, frame_size=(W, H), T0=d.get_T(0))
    d.add_class("c", descs, feats)
    for i in range(NB):
        d.upload_frame(i, *frames[i])
    d.set_tuning(lm.TUNE_SCAN_FORM, 3)
    d.match_batch(NB, 80.0, cap_per_frame=4096)
    out = []
    for n in sizes:
        out.append("%d:%.1f" % (n, min(d.time_scan_batch(0, n, 80.0, iters=20, variant=0) for _ in range(3))))
    print("R=%s  " % os.environ.get("LM_SCANL_R", "auto") + "  ".join(out), flush=True)
    d.close()
else:
    sizes = sys.argv[1:] or ["16", "32", "48", "64", "96", "128"]
    for R in ("auto", "2", "3", "4", "5", "6", "8", "10", "12", "16", "24", "32"):
        env = dict(os.environ)
        if R != "auto":
            env["LM_SCANL_R"] = R
        subprocess.run([sys.executable, os.path.abspath(__file__), "child"] + sizes, env=env)
