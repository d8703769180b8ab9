#!/bin/bash
# One call of the slow-step outlier hunt: the default step, SlowFast on the launch stream, no SlowFast at all, twice each, interleaved.
# usage (GPU box): bash scripts/outlier_probe.sh TAG
tag=${1:-x}
for i in 1 2; do
  python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-parity --no-decode > gpurun_out/op_${tag}_d$i.json 2>/dev/null
  python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-parity --no-decode --serial-motion > gpurun_out/op_${tag}_s$i.json 2>/dev/null
  python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-parity --no-decode --motion input > gpurun_out/op_${tag}_i$i.json 2>/dev/null
done
python - <<PY
import json
for f in ("d1","s1","i1","d2","s2","i2"):
    try:
        d=json.loads(open("gpurun_out/op_${tag}_%s.json"%f).read().strip().splitlines()[-1])
        print(f, round(d["ms_per_step"],1), d["roofline"]["measured"][-38:], "gemm", round(d["roofline"]["gemm_ms_per_step"],1))
    except Exception as e: print(f, "ERR", e)
PY
