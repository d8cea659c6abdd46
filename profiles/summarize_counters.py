"""Reduces the rocprofv3 passes of tools/collect_counters.sh for ONE bench command (one lane, one launch per kernel) to one
JSON file that bench.py reads: per kernel the clean duration (kernel stats), HBM-side traffic (FETCH_SIZE / WRITE_SIZE), L1 -> L2
requests and L2 hits (TCP_* / TCC_*), vector-ALU activity (SQ_*), and the command's identity (`meta`) so that bench.py uses the
file ONLY for the command and the kernel sources it was collected with (a stale file must not describe another workload).

  python profiles/summarize_counters.py <out.json> <bench json line of the one-lane run> <kernel-stats dir> \
         [F=<dir of --pmc FETCH_SIZE>] [W=<dir WRITE_SIZE>] [TCP=<dir>] [TCC=<dir>] [SQ=<dir>] [LDS=<dir>] [TCPNP=<dir TCP pass of --no-prune>] \
         [NPJSON=<bench json line of the --no-prune run>]

Units / corrections (MI355X_MICROARCH.md, "HBM" and "rocprofv3 PMC"): FETCH_SIZE and WRITE_SIZE are in KB (x 1024); on gfx950
FETCH_SIZE reports exactly 1/2 of the bytes of a wide coalesced streaming read (16 B per lane), so it is doubled; WRITE_SIZE is
exact for 16-B-per-lane stores.  Separate --pmc passes (FETCH_SIZE takes 3 of the 4 TCC slots, WRITE_SIZE 2).  SQ_ACTIVE_INST_* and
SQ_WAVE_CYCLES count quad-cycles (x 4 = cycles).  Per dispatch the rows of all instances (XCDs / SEs) are SUMMED; `instances`
says how many rows a counter had per dispatch, so GRBM_GUI_ACTIVE / instances = the cycles one XCD was active.  The first two
launches of every kernel (warm-up) are skipped.  A TCP -> TCC read request is calibrated on the exhaustive scan, whose requested
bytes are known exactly (16 B per lane and feature): 128 B if requested bytes / request > 70, else 64 B."""
import collections
import csv
import glob
import json
import sys


def short(n):
    n = n.replace("(anonymous namespace)::", "")
    if n.startswith("void "):
        n = n[5:]
    return n.split("(")[0]


def kernel_stats(path):
    out = {}
    for f in glob.glob(path + "/**/*kernel_stats.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            out[short(r["Name"])] = {"calls": int(r["Calls"]), "avg_us": round(float(r["AverageNs"]) / 1e3, 2),
                                     "min_us": round(float(r["MinNs"]) / 1e3, 2), "max_us": round(float(r["MaxNs"]) / 1e3, 2)}
    return out


def counters(path):
    """kernel -> counter -> (average over launches of the per-dispatch sum, rows per dispatch, launches)"""
    tab = collections.defaultdict(lambda: collections.defaultdict(list))
    rows = collections.defaultdict(lambda: collections.defaultdict(int))
    for f in glob.glob(path + "/**/*counter_collection.csv", recursive=True):
        per, nrow, name = collections.defaultdict(float), collections.defaultdict(int), {}
        for r in csv.DictReader(open(f)):
            key = (int(r["Dispatch_Id"]), r["Counter_Name"])
            per[key] += float(r["Counter_Value"]); nrow[key] += 1
            name[int(r["Dispatch_Id"])] = short(r["Kernel_Name"])
        for (disp, c), v in sorted(per.items()):
            tab[name[disp]][c].append(v)
            rows[name[disp]][c] = max(rows[name[disp]][c], nrow[(disp, c)])
    out = {}
    for k, cs in tab.items():
        out[k] = {}
        for c, v in cs.items():
            use = v[2:] if len(v) > 2 else v
            out[k][c] = {"per_launch": sum(use) / len(use), "instances": rows[k][c], "launches": len(v)}
    return out


def main():
    out_path, bench_json, stats_dir = sys.argv[1], sys.argv[2], sys.argv[3]
    opt = dict(a.split("=", 1) for a in sys.argv[4:])
    line = [l for l in open(bench_json).read().splitlines() if l.startswith('{"metric"')][-1]
    bj = json.loads(line)
    res = {"meta": bj.get("counters_meta"), "bench_value": bj.get("value"),
           "stage_us_per_frame_one_lane": bj["roofline"].get("stage_us_per_frame_one_lane"),
           "frames_per_launch": bj["roofline"]["frames_per_launch"], "baseline_config": bj["config"]["baseline_config"],
           "kernels": {}, "passes": sorted(opt)}
    ks = kernel_stats(stats_dir)
    for k, v in ks.items():
        res["kernels"][k] = dict(v)
    passes = {p: counters(opt[p]) for p in ("F", "W", "TCP", "TCC", "SQ", "LDS") if p in opt}      # (LDS, r06: SQ_INSTS_LDS, SQ_LDS_IDX_ACTIVE, SQ_LDS_BANK_CONFLICT -- k_scanl reads its planes from LDS)
    for p, tab in passes.items():
        for k, cs in tab.items():
            e = res["kernels"].setdefault(k, {})
            for c, v in cs.items():
                e[c] = round(v["per_launch"], 1)
                if c in ("GRBM_GUI_ACTIVE", "SQ_BUSY_CYCLES"):
                    e[c + "_instances"] = v["instances"]
    for k, e in res["kernels"].items():
        if "FETCH_SIZE" in e or "WRITE_SIZE" in e:
            rd, wr = 2.0 * e.get("FETCH_SIZE", 0.0) * 1024.0, e.get("WRITE_SIZE", 0.0) * 1024.0
            e["hbm_read_bytes_per_launch"], e["hbm_write_bytes_per_launch"] = round(rd), round(wr)
            e["hbm_bytes_per_launch"] = round(rd + wr)
    res["bytes_per_request"] = 128
    if "TCPNP" in opt and "NPJSON" in opt:
        try:
            npl = [l for l in open(opt["NPJSON"]).read().splitlines() if l.startswith('{"metric"')][-1]
            known = float(json.loads(npl)["roofline"]["load_bytes_per_launch"])
            npr = counters(opt["TCPNP"])
            scan = [k for k in npr if k.startswith("k_scan4")][0]
            req = npr[scan]["TCP_TCC_READ_REQ_sum"]["per_launch"]
            res["calibration"] = {"kernel": scan + " (--no-prune: every feature loaded)", "requested_bytes_per_launch": known,
                                  "TCP_TCC_READ_REQ_per_launch": round(req, 1), "requested_bytes_per_request": round(known / req, 2),
                                  "conclusion": "a TCP -> TCC read request is one 128-B line" if known / req > 70 else "a TCP -> TCC read request is 64 B"}
            res["bytes_per_request"] = 128 if known / req > 70 else 64
        except Exception as e:   # noqa: BLE001
            res["calibration"] = {"error": "%s: %s" % (type(e).__name__, e)}
    json.dump(res, open(out_path, "w"), indent=1, sort_keys=True)
    print("%-28s %6s %9s %11s %11s %12s" % ("kernel", "calls", "avg us", "HBM MB", "L2 req M", "VALU inst M"))
    for k, e in sorted(res["kernels"].items(), key=lambda kv: -kv[1].get("avg_us", 0) * kv[1].get("calls", 0)):
        print("%-28s %6s %9s %11s %11s %12s" % (k[:28], e.get("calls", "-"), e.get("avg_us", "-"),
                                                 round(e["hbm_bytes_per_launch"] / 1e6, 1) if "hbm_bytes_per_launch" in e else "-",
                                                 round(e["TCP_TCC_READ_REQ_sum"] / 1e6, 2) if "TCP_TCC_READ_REQ_sum" in e else "-",
                                                 round(e["SQ_INSTS_VALU"] / 1e6, 2) if "SQ_INSTS_VALU" in e else "-"))
    print("calibration", res.get("calibration"))


if __name__ == "__main__":
    main()
