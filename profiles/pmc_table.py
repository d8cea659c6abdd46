"""Per-kernel averages of every counter in rocprofv3 --pmc passes (counter_collection.csv).
usage: python profiles/pmc_table.py <dir> [<dir> ...] [--kernel k_scan]"""
import collections, csv, glob, sys


def main():
    dirs = [a for a in sys.argv[1:] if not a.startswith("--")]
    kfilter = sys.argv[sys.argv.index("--kernel") + 1] if "--kernel" in sys.argv else None
    tab = collections.defaultdict(lambda: collections.defaultdict(list))
    for d in dirs:
        for f in glob.glob(d + "/*/*counter_collection.csv"):
            per = collections.defaultdict(float)
            name = {}
            for r in csv.DictReader(open(f)):
                key = (r["Dispatch_Id"], r["Counter_Name"])
                per[key] += float(r["Counter_Value"])
                n = r["Kernel_Name"]
                name[r["Dispatch_Id"]] = n.split("::")[-1].split("(")[0] if "::" in n else n.split("(")[0]
            for (disp, c), v in per.items():
                tab[name[disp]][c].append(v)
    for k in sorted(tab):
        if kfilter and kfilter not in k:
            continue
        print(k)
        for c in sorted(tab[k]):
            v = tab[k][c][2:] or tab[k][c]
            print("   %-36s %16.1f  (n=%d)" % (c, sum(v) / len(v), len(v)))


if __name__ == "__main__":
    main()
