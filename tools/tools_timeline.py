"""Scratch: run under `rocprofv3 --kernel-trace --output-format csv -d DIR -- python3 tools/tools_timeline.py`; then
`python3 tools/tools_timeline.py --summarize DIR` prints the kernel timeline of one steady-state single-frame match."""
import importlib, sys, os, glob, csv
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 2 and sys.argv[1] == "--summarize":
    f = glob.glob(os.path.join(sys.argv[2], "**", "*kernel_trace.csv"), recursive=True)[0]
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    ends = [i for i, r in enumerate(rows) if "k_sort_unique" in r["Kernel_Name"]]
    lo, hi = ends[-3] + 1, ends[-2] + 1           # the second to last sequence
    t0 = int(rows[lo]["Start_Timestamp"])
    for r in rows[lo:hi]:
        name = r["Kernel_Name"].split("(")[0].replace("(anonymous namespace)::", "").replace("void ", "")
        print("%-28s grid %7s  start %7.1f  duration %6.1f" % (name[:28], r["Grid_Size_X"] if "Grid_Size_X" in r else r.get("Grid_Size", "?"),
              (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
    print("span %.1f us" % ((int(rows[hi - 1]["End_Timestamp"]) - t0) / 1e3))
    sys.exit(0)
lm = importlib.import_module("line-mod-pipeline_amd")
synth = importlib.import_module("line-mod-pipeline_amd.synth")
W, H = 640, 480
CO = os.environ.get("LM_COLOR_ONLY") == "1"      # r05: the reference's shipped modality (bench.py's `latency` block): colour only, 1950 templates of variable size
M = 1 if CO else 2
d = lm.Detector(lm.default_config(color_only=CO, width=W, height=H, frame_slots=8))
frames = [synth.make_frame(W, H, seed=1234 + i) for i in range(8)]
d.upload_frame(0, frames[0][0], None if CO else frames[0][1]); d.prepare_slot(0)
q = {(l, m): d.debug_read(0, 0, l, m).reshape(H >> l, W >> l) for l in range(2) for m in range(M)}
if CO:
    descs, feats, _ = synth.make_bank(1950, 1, 2, seed=4321, quantized=q, crop_fraction=0.1, frame_size=(W, H), T0=d.get_T(0))
else:
    descs, feats, _ = synth.make_bank(3000, 2, 2, seed=4321, fixed_l0_size=(96, 96), quantized=q, crop_fraction=0.1,
                                      frame_size=(W, H), T0=d.get_T(0))
d.add_class("c", descs, feats)
for i, (b, dp) in enumerate(frames):
    d.upload_frame(i, b, None if CO else dp)
if os.environ.get("LM_PHASES") is not None:
    d.set_tuning(lm.TUNE_PHASE_MAX_SLOTS, int(os.environ["LM_PHASES"]))
for k in range(40):
    d.match_slot(1 + k % 7, 80.0, 0)
d.close()
