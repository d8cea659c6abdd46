"""Scratch: sums SQ_INSTS_VALU / SQ_ACTIVE_INST_VALU / SQ_BUSY_CYCLES per kernel from a rocprofv3 --pmc pass
(one-lane bench run) and prints vector instructions per frame.  usage: valu_util.py <dir> <frames_per_launch>"""
import collections, csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
frames = int(sys.argv[2])
d = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"]
    n = n.split("::")[-1].split("(")[0] if "::" in n else n.split("(")[0]
    d[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
tot = collections.Counter()
print("%-28s %8s %14s %14s" % ("kernel", "launches", "INSTS_VALU/frame", "ACTIVE_INST_VALU/frame"))
for k, c in sorted(d.items(), key=lambda kv: -sum(kv[1].get("SQ_INSTS_VALU", [0]))):
    iv = c.get("SQ_INSTS_VALU", [0]); av = c.get("SQ_ACTIVE_INST_VALU", [0])
    n = len(iv)
    per = sum(iv) / max(n, 1) / frames
    # kernels launched once per level / modality: count launches per lane-step by the scan's launches
    print("%-28s %8d %14.0f %14.0f" % (k[:28], n, per, sum(av) / max(n, 1) / frames))
scan_n = len(d.get("k_scan4", {}).get("SQ_INSTS_VALU", [1]))
for k, c in d.items():
    tot["valu"] += sum(c.get("SQ_INSTS_VALU", [0])) / scan_n / frames
    tot["active"] += sum(c.get("SQ_ACTIVE_INST_VALU", [0])) / scan_n / frames
print("all kernels, per frame: %.0f vector wave-instructions, %.0f ACTIVE_INST_VALU (quad-cycles)" % (tot["valu"], tot["active"]))
