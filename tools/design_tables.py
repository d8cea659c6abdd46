"""Prints the per-kernel tables of DESIGN.md section 7 from profiles/r04_counters_c{2,3,5}.json (usage: python tools/design_tables.py [tag])."""
import json
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
for c, title in ((2, "config 2, 96 frames per launch"), (3, "config 3, 128 frames per launch"), (5, "config 5, 8 frames per launch")):
    d = json.load(open("profiles/%s_counters_c%d.json" % (tag, c)))
    ks = d["kernels"]
    scan = [v for k, v in ks.items() if k.startswith("k_scan") and not k.startswith("k_scan1_exact")][0]
    n = scan["calls"]
    clock = scan["GRBM_GUI_ACTIVE"] / 8.0 / scan["avg_us"] / 1e3
    print("| %s (clock %.2f GHz) | µs per launch | launches | VALU instr (M) | VALU busy | HBM TB/s | L2 frac |\n|---|---|---|---|---|---|---|" % (title, clock))
    tot = 0.0
    for k, v in sorted(ks.items(), key=lambda kv: -kv[1].get("avg_us", 0) * kv[1].get("calls", 0)):
        if v.get("calls", 0) < 8 or k.startswith("__amd") or "avg_us" not in v:
            continue
        us, per = v["avg_us"], v["calls"] / n
        valu = v.get("SQ_ACTIVE_INST_VALU", 0) * 4 / 1024 / (us * clock * 1e3)
        tot += v.get("SQ_INSTS_VALU", 0) * per
        print("| `%s` | %.1f | %.0f | %.1f | %.2f | %.2f | %.2f |" % (k, us, per, v.get("SQ_INSTS_VALU", 0) * per / 1e6, valu, v.get("hbm_bytes_per_launch", 0) / us / 1e6,
                                                                        v.get("TCP_TCC_READ_REQ_sum", 0) * 128 / us / 1e6 / 34.5))
    pre = sum(v.get("hbm_bytes_per_launch", 0) * v["calls"] / n for k, v in ks.items()
              if v.get("calls", 0) >= 8 and k.startswith(("k_blur", "k_cblur", "k_cgrad", "k_dnormal", "k_dmedian", "k_lm_", "k_pyrdown")))
    print("\nVALU wave-instructions per launch: %.1f M; HBM-side traffic of a3-a10: %.2f MB per frame\n" % (tot / 1e6, pre / d["frames_per_launch"] / 1e6))
