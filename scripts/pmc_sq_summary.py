"""Per-kernel means of SQ counters from rocprofv3 --pmc passes: python scripts/pmc_sq_summary.py <kernel substring> dir1 [dir2 ...]"""
import collections, csv, glob, sys
pat = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for d in sys.argv[2:]:
    for f in glob.glob(d + "/**/*_counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            n = r["Kernel_Name"]
            if pat in n:
                key = n.split("::")[-1].split("(")[0]
                a = agg[key][r["Counter_Name"]]
                a[0] += 1; a[1] += float(r["Counter_Value"])
for k in sorted(agg):
    print(k)
    for cname in sorted(agg[k]):
        n, v = agg[k][cname]
        print(f"   {cname:32s} {v / n:16.0f}   ({n} launches)")
