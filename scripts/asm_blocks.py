"""Instruction mix of the basic blocks of one kernel that hold MFMAs (hipcc -S output).

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -S --cuda-device-only [-fno-slp-vectorize] -o /tmp/k.s file.hip
    python scripts/asm_blocks.py /tmp/k.s <mangled-name-substring> [min_mfma]
"""
import collections
import re
import sys

s = open(sys.argv[1]).read().split("\n")
key = sys.argv[2]
min_mfma = int(sys.argv[3]) if len(sys.argv) > 3 else 8
start = next(i for i, l in enumerate(s) if l.startswith("_ZN") and key in l and l.rstrip().split(":")[0].endswith("AttnArgs") | True and l.split(":")[0].find(key) >= 0)
end = next(i for i in range(start, len(s)) if "s_endpgm" in s[i])
blocks, cur, name = [], [], "entry"
for l in s[start + 1:end]:
    l = l.strip()
    if re.match(r"^\.LBB\d+_\d+:", l):
        blocks.append((name, cur)); name, cur = l, []
    elif l and not l.startswith(";") and not l.startswith("."):
        cur.append(l.split()[0])
blocks.append((name, cur))
for name, b in blocks:
    c = collections.Counter(b)
    nm = sum(v for k, v in c.items() if "mfma" in k)
    if nm >= min_mfma:
        valu = {k: v for k, v in c.items() if k.startswith("v_") and "mfma" not in k}
        print(f"{name} instr {len(b)} mfma {nm} valu {sum(valu.values())} ds {sum(v for k, v in c.items() if k.startswith('ds_'))} "
              f"salu {sum(v for k, v in c.items() if k.startswith('s_'))}")
        print("    ", sorted(((v, k) for k, v in valu.items()), reverse=True)[:24])
