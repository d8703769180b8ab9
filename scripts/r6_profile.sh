#!/bin/bash
# Round-6 evidence run on one MI355X box (one gpurun call): rocprofv3 kernel-trace stats of the bench command, the two PMC passes behind
# roofline.traffic (separate runs, as MI355X_MICROARCH.md prescribes), the one-rank RCCL variant of the bench.  Outputs under gpurun_out/.
set -o pipefail
export TMPDIR=/tmp
B="--no-cpu-baseline --no-decode --no-prof --no-parity"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r6_prof -- python3 bench.py --steps 10 --warmup 2 $B > gpurun_out/r6_prof_bench.json 2> gpurun_out/r6_prof_bench.err &&
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/r6_pmc_fetch -- python3 bench.py --steps 3 --warmup 1 --no-graph --no-settle $B > gpurun_out/r6_pmc_fetch.json 2> gpurun_out/r6_pmc_fetch.err &&
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/r6_pmc_write -- python3 bench.py --steps 3 --warmup 1 --no-graph --no-settle $B > gpurun_out/r6_pmc_write.json 2> gpurun_out/r6_pmc_write.err &&
python3 scripts/pmc_summary.py gpurun_out/r6_pmc_fetch gpurun_out/r6_pmc_write gpurun_out/r6_gemm_traffic.json &&
python3 bench.py --force-dp --steps 10 --warmup 3 --no-cpu-baseline --no-parity --no-decode > gpurun_out/r6_bench_force_dp.json 2> gpurun_out/r6_bench_force_dp.err
