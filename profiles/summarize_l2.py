"""Reduces the L1 / L2 counter passes of tools/collect_counters.sh to per-kernel averages per launch and calibrates the
size of a TCP -> TCC read request on the exhaustive scan, whose requested bytes are known exactly.

  python profiles/summarize_l2.py <dir TCP pass> <dir TCC pass> <dir TCP pass of --no-prune> <bench json of --no-prune>
                                  <frames_per_launch> <BASELINE config> <out.json>

Counters: TCP_TCC_READ_REQ_sum (read requests the vector L1s send to the L2s), TCP_TOTAL_CACHE_ACCESSES_sum (L1 tag
lookups), TCC_REQ_sum / TCC_HIT_sum / TCC_MISS_sum (L2 requests, hits, misses).  Per dispatch the rows of all XCDs / SEs
are summed.  Bytes per request: the exhaustive k_scan4 requests exactly `load_bytes_per_launch` bytes (bench.py:
16 B per lane and feature, from lm_scan_load_bytes); half-wave loads of 512 contiguous, dword-aligned bytes touch 4 or 5
128-byte lines, so requests x 128 B / requested bytes is expected between 1.0 and 1.25 if a request is one 128-B line, and
about half that if it were 64 B."""
import collections
import csv
import glob
import json
import sys


def per_kernel(path):
    tab = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(path + "/**/*counter_collection.csv", recursive=True):
        per, name = collections.defaultdict(float), {}
        for r in csv.DictReader(open(f)):
            per[(r["Dispatch_Id"], r["Counter_Name"])] += float(r["Counter_Value"])
            n = r["Kernel_Name"]
            name[r["Dispatch_Id"]] = n.split("::")[-1].split("(")[0] if "::" in n else n.split("(")[0]
        for (disp, c), v in sorted(per.items(), key=lambda kv: int(kv[0][0])):
            tab[name[disp]][c].append(v)
    out = {}
    for k, cs in tab.items():
        out[k] = {c: (sum(v[2:]) / len(v[2:]) if len(v) > 2 else sum(v) / len(v)) for c, v in cs.items()}   # skip warm-up launches
        out[k]["launches"] = max(len(v) for v in cs.values())
    return out


def main():
    d_tcp, d_tcc, d_np, j_np, frames, config, out = sys.argv[1], sys.argv[2], sys.argv[3], sys.argv[4], int(sys.argv[5]), int(sys.argv[6]), sys.argv[7]
    tcp, tcc, npr = per_kernel(d_tcp), per_kernel(d_tcc), per_kernel(d_np)
    res = {"frames_per_launch": frames, "baseline_config": config, "kernels": {}}
    for k in sorted(set(tcp) | set(tcc)):
        e = {}
        e.update({c: round(v, 1) for c, v in tcp.get(k, {}).items()})
        e.update({c: round(v, 1) for c, v in tcc.get(k, {}).items()})
        res["kernels"][k] = e
    try:
        bj = json.load(open(j_np))
        known = float(bj["roofline"]["load_bytes_per_launch"])
        scan = [k for k in npr if k.startswith("k_scan4")]
        req = npr[scan[0]]["TCP_TCC_READ_REQ_sum"]
        res["calibration"] = {"kernel": scan[0] + " (--no-prune: every feature loaded)", "requested_bytes_per_launch": known,
                              "TCP_TCC_READ_REQ_per_launch": round(req, 1), "requested_bytes_per_request": round(known / req, 2),
                              "TCP_TOTAL_CACHE_ACCESSES_per_launch": round(npr[scan[0]].get("TCP_TOTAL_CACHE_ACCESSES_sum", 0.0), 1),
                              "conclusion": "a TCP -> TCC read request is one 128-B line" if known / req > 70 else "a TCP -> TCC read request is 64 B"}
        res["bytes_per_request"] = 128 if known / req > 70 else 64
    except Exception as e:   # noqa: BLE001
        res["calibration"] = {"error": "%s: %s" % (type(e).__name__, e)}
        res["bytes_per_request"] = 128
    json.dump(res, open(out, "w"), indent=1, sort_keys=True)
    for k, v in sorted(res["kernels"].items()):
        print(k, v)
    print("calibration", res["calibration"])


if __name__ == "__main__":
    main()
