#!/bin/bash
# Round-6 SQ / GRBM counters of the GEMM and attention launches INSIDE the scorer's step: one kernel-trace run (durations) and one --pmc pass (counters) of the same
# command on the same box; summaries by scripts/pmc_sq_summary.py + pmc_sq_derive.py.  Separate runs, as gpurun requires (--pmc never with a trace).
set -o pipefail
export TMPDIR=/tmp
B="--steps 2 --warmup 1 --no-cpu-baseline --no-decode --no-prof --no-parity --no-settle --no-graph"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r6_sq_trace -- python3 bench.py $B > gpurun_out/r6_sq_trace.json 2> gpurun_out/r6_sq_trace.err &&
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/r6_sq_pmc -- python3 bench.py $B > gpurun_out/r6_sq_pmc.json 2> gpurun_out/r6_sq_pmc.err &&
STATS=$(find gpurun_out/r6_sq_trace -name "*kernel_stats.csv" | head -1) &&
python3 scripts/pmc_sq_summary.py gemm256_kernel gpurun_out/r6_sq_pmc > gpurun_out/r6_gemm_sq_raw.txt && python3 scripts/pmc_sq_derive.py gpurun_out/r6_gemm_sq_raw.txt $STATS > gpurun_out/r6_gemm_sq.txt &&
python3 scripts/pmc_sq_summary.py attn_fwd_kernel gpurun_out/r6_sq_pmc > gpurun_out/r6_attn_sq_raw.txt && python3 scripts/pmc_sq_derive.py gpurun_out/r6_attn_sq_raw.txt $STATS > gpurun_out/r6_attn_sq.txt
