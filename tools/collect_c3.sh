#!/bin/bash
# GPU box: the evidence kept under profiles/ for BASELINE configs[2] (bench.py --workload c3: 2048x1536 frames, 0.5 mm voxels):
# the bench line, rocprofv3 kernel stats of one timed pass, and the same --pmc passes as tools/collect_profiles.sh.
# Summarise: tools/pmc_table.py gpurun_out/<tag>_mem k_integrate k_update k_replay k_register k_depinc > ...
# usage: tools/collect_c3.sh <tag>
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-c3}
mkdir -p $R/gpurun_out
cd $R
timeout -k 10 400 python3 bench.py --workload c3 --cpu-sample 0 --host-path-frames 0 > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err || { echo "bench failed"; exit 1; }
echo "bench done"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_stats -o s -- python3 $R/bench.py --workload c3 --repeats 1 --warmup 0 --cpu-sample 0 --host-path-frames 0 > $R/gpurun_out/${TAG}_stats.json 2> $R/gpurun_out/${TAG}_stats.err || { echo "stats failed"; exit 1; }
echo "stats done"
cd $R
export BENCH_ARGS="--workload c3"
tools/pmc_passes.sh ${TAG}_mem "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_ATOMIC_sum" "TCC_HIT_sum TCC_MISS_sum" || exit 1
tools/pmc_passes.sh ${TAG}_sq "GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
  "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" || exit 1
python3 tools/pmc_table.py gpurun_out/${TAG}_mem k_integrate k_update k_replay k_register k_depinc k_normal k_buffer > gpurun_out/${TAG}_mem_counters.md
python3 tools/pmc_table.py gpurun_out/${TAG}_sq k_integrate k_update k_replay k_register k_depinc k_normal k_buffer > gpurun_out/${TAG}_sq_counters.md
python3 tools/pmc_summary.py gpurun_out/${TAG}_mem gpurun_out/${TAG}_pmc_hot_path gpurun_out/${TAG}_sq --workload c3 > /dev/null
echo "c3 profiles done"
