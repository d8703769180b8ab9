"""Per-kernel register / scratch / occupancy table of one HIP source (hipcc -Rpass-analysis=kernel-resource-usage), gfx950.

    python scripts/kernel_resources.py aigv-assessor_amd/csrc/gemm256.hip [extra hipcc flags]
"""
import re
import subprocess
import sys

src = sys.argv[1]
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value", "-c", src, "-o", "/dev/null",
       "-Rpass-analysis=kernel-resource-usage"] + sys.argv[2:]
err = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = None
rows = {}
for line in err.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip() or m.group(1)
        rows[cur] = {}
        continue
    m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)", line)
    if m and cur:
        rows[cur][m.group(1).strip()] = int(m.group(2))
for k, v in rows.items():
    print(f"{k[:110]:110s} VGPR {v.get('VGPRs', -1):4d} AGPR {v.get('AGPRs', -1):4d} spill {v.get('VGPRs Spill', -1):3d} scratch {v.get('ScratchSize', -1):4d} "
          f"occ {v.get('Occupancy', -1)} LDS {v.get('LDS Size', -1)}")
