"""Reduces two rocprofv3 counter passes (FETCH_SIZE, WRITE_SIZE; separate --pmc runs of the same
bench.py command, as MI355X_MICROARCH.md section 'HBM' prescribes) to per-kernel HBM traffic per launch.

  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d <dirF> -- python3 bench.py --steps 20 --warmup 2 --no-cpu-baseline
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d <dirW> -- python3 bench.py --steps 20 --warmup 2 --no-cpu-baseline
  python profiles/summarize_pmc.py <dirF> <dirW> <frames_per_launch> <out.json> [<BASELINE config: 2 | 3>]

Units/corrections (guide): both counters are in KB (x1024); on gfx950 FETCH_SIZE reports exactly 1/2 of
the bytes of a wide coalesced streaming read (16 B/lane), so it is doubled; WRITE_SIZE is exact for
16-B-per-lane stores.  Other access widths are uncalibrated: treat non-scan rows as indicative.
"""
import collections
import csv
import glob
import json
import sys


def agg(path, cname):
    f = glob.glob(path + "/*/*counter_collection.csv")[0]
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != cname:
            continue
        n = r["Kernel_Name"]
        n = n.split("::")[-1].split("(")[0] if "::" in n else n
        d[n].append((float(r["Counter_Value"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
    return d


def main():
    dir_f, dir_w, frames, out = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
    config = int(sys.argv[5]) if len(sys.argv) > 5 else 2
    fe, wr = agg(dir_f, "FETCH_SIZE"), agg(dir_w, "WRITE_SIZE")
    res = {"frames_per_launch": frames, "baseline_config": config, "kernels": {}}
    for k in fe:
        v = [x for x, _ in fe[k]][2:]           # skip the warm-up launches
        w = [x for x, _ in wr.get(k, [])][2:]
        t = [t for _, t in fe[k]][2:]
        if not v:
            continue
        fetch_kb, write_kb = sum(v) / len(v), (sum(w) / len(w)) if w else 0.0
        res["kernels"][k] = {
            "launches": len(v), "FETCH_SIZE_KB_raw": round(fetch_kb, 1), "WRITE_SIZE_KB": round(write_kb, 1),
            "hbm_bytes_per_launch": round((2.0 * fetch_kb + write_kb) * 1024.0),
            "avg_us_profiled": round(sum(t) / len(t) / 1000.0, 1)}
    json.dump(res, open(out, "w"), indent=1, sort_keys=True)
    for k, v in sorted(res["kernels"].items()):
        print("%-34s %12.2f MB/launch" % (k, v["hbm_bytes_per_launch"] / 1e6))


if __name__ == "__main__":
    main()
