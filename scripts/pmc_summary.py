"""Summarise rocprofv3 PMC passes (FETCH_SIZE / WRITE_SIZE, collected in SEPARATE runs) for the GEMM kernels.

    python scripts/pmc_summary.py gpurun_out/pmc_fetch gpurun_out/pmc_write profiles/r1_gemm_traffic.json

Units and corrections as MI355X_MICROARCH.md §HBM prescribes: both counters are in KiB; on gfx950 FETCH_SIZE reports half
of the bytes of wide coalesced streaming reads (16 B/lane, global_load and LDS-DMA alike) -> doubled; WRITE_SIZE is exact
for 16-B stores (the epilogue here stores 8 B per lane: taken as reported).  The counters sit on the L2's fabric side, so
Infinity-Cache hits are included: this is L2<->fabric traffic, an upper bound on HBM traffic.
"""
import collections, csv, glob, json, os, sys

def load(d):
    f = max(glob.glob(d + "/*/*_counter_collection.csv"), key=os.path.getmtime)   # the latest pass if the directory holds several
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "gemm256_kernel" in n or "gemm_bf16_kernel" in n:
            key = n.split("::")[-1].split("(")[0]
            agg[key][0] += 1
            agg[key][1] += float(r["Counter_Value"])
    return agg

fetch, write = load(sys.argv[1]), load(sys.argv[2])
out = {"kernels": {}, "note": "bytes per launch; fetch = FETCH_SIZE KiB * 1024 * 2 (gfx950 correction), write = WRITE_SIZE KiB * 1024"}
tl = tf = tw = 0
for k in sorted(fetch):
    n = fetch[k][0]
    fb = fetch[k][1] * 1024 * 2 / n
    wb = write.get(k, [1, 0.0])[1] * 1024 / max(1, write.get(k, [1, 0.0])[0])
    out["kernels"][k] = {"launches": n, "fetch_bytes_per_launch": fb, "write_bytes_per_launch": wb}
    tl += n; tf += fb * n; tw += wb * n
out["all_gemm"] = {"launches": tl, "traffic_bytes_per_launch": (tf + tw) / tl, "fetch_bytes_per_launch": tf / tl, "write_bytes_per_launch": tw / tl}
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out["all_gemm"]))
