"""Run the bare LDS-fed MFMA loop (scripts/mfma_lds_probe.hip, built as scripts/_abl/mfma_lds_probe) on this MI355X and write what bench.py
quotes as `roofline.sustained_peak`: the median TFLOP/s of the shipped wave tile (128x64, two waves per SIMD) - ds_read_b128 +
v_mfma_f32_16x16x32_bf16 and nothing else, at the clock the chip holds under that load.

    python scripts/mfma_lds_probe_summary.py profiles/r6_mfma_lds_probe      # -> .txt (the probe's lines) and .json
"""
import json
import re
import subprocess
import sys
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = subprocess.run([os.path.join(ROOT, "scripts", "_abl", "mfma_lds_probe")], capture_output=True, text=True, check=True).stdout
rows = [(m.group(1).strip(), float(m.group(2)), float(m.group(3)), float(m.group(4)))
        for m in re.finditer(r"^(.*?)\s+([\d.]+) ms\s+([\d.]+) TFLOP/s\s+in-kernel clock\s+([\d.]+) MHz", out, re.M)]
shipped = sorted(r for r in rows if r[0].startswith("128x64"))
assert shipped, out
med = shipped[len(shipped) // 2]
base = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "mfma_lds_probe")
with open(base + ".txt", "w") as f:
    f.write("# scripts/mfma_lds_probe.hip on an MI355X (round 6 re-measurement): the GEMM inner loop alone - ds_read_b128 + v_mfma_f32_16x16x32_bf16 on random bf16\n"
            "# operands in LDS, no DMA / barriers / epilogue; 256 workgroups, wall time by HIP events, clock = d s_memtime / d s_memrealtime x 100 MHz.\n" + out)
json.dump({"sustained_peak_tflops": med[2], "in_kernel_clock_mhz": med[3], "runs": [r[2] for r in shipped], "wave_tile": med[0],
           "what": "bare LDS-fed MFMA loop of the shipped 128x64 wave tile (scripts/mfma_lds_probe.hip): the ceiling a bf16 GEMM kernel of this tiling can reach "
                   "on this chip at the clock it holds under MFMA load; spec peak 2500 TFLOP/s assumes 2.4 GHz"}, open(base + ".json", "w"), indent=1)
print(open(base + ".json").read())
