"""Per-launch times of one SlowFast features() call from a rocprofv3 kernel trace:
    cd /tmp && rocprofv3 --kernel-trace --output-format csv -d OUT -- python3 scripts/slowfast_bench.py 4
    python scripts/slowfast_layers.py OUT/**/*kernel_trace.csv [launches per call = auto]
Prints the launches of the LAST call in order (kernel, grid, workgroups, us) and the sums per kernel."""
import csv, glob, sys
from collections import defaultdict

paths = [p for a in sys.argv[1:] if not a.isdigit() for p in glob.glob(a, recursive=True)]
rows = []
for p in paths:
    with open(p) as f:
        for r in csv.DictReader(f):
            rows.append(r)
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
# one call = the span between two sf_repack launches
starts = [i for i, n in enumerate(names) if "sf_repack" in n]
if len(starts) < 2:
    sys.exit("no repack launches found")
lo, hi = starts[-2], starts[-1]
tot = 0.0
per = defaultdict(lambda: [0, 0.0])
t_first, t_last = int(rows[lo]["Start_Timestamp"]), int(rows[hi - 1]["End_Timestamp"])
for r in rows[lo:hi]:
    us = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    g = [int(r[k]) for k in ("Grid_Size_X", "Grid_Size_Y", "Grid_Size_Z")]
    w = [int(r[k]) for k in ("Workgroup_Size_X", "Workgroup_Size_Y", "Workgroup_Size_Z")]
    wg = (g[0] // w[0]) * (g[1] // w[1]) * (g[2] // w[2])
    short = r["Kernel_Name"].replace("void ", "", 1).replace("(anonymous namespace)::", "").split("(")[0]
    print(f"{short:34s} wgs {wg:6d} ({g[0] // w[0]} x {g[1] // w[1]} x {g[2] // w[2]})  {us:8.1f} us")
    tot += us
    per[short][0] += 1
    per[short][1] += us
print(f"-- {hi - lo} launches, kernel time {tot:.1f} us, wall {(t_last - t_first) / 1e3:.1f} us")
for k, (n, us) in sorted(per.items(), key=lambda kv: -kv[1][1]):
    print(f"   {k:34s} {n:4d} launches {us:8.1f} us")
