#!/bin/bash
# Runs on the GPU box (gpurun): the evidence kept under profiles/ for one build.
#   1. the default bench line                                   -> gpurun_out/<tag>_bench.json
#   2. rocprofv3 --kernel-trace --stats of one timed pass       -> gpurun_out/<tag>_stats/
#   3. rocprofv3 --pmc passes, ONE counter set per pass (never combined with other trace domains):
#        memory side (FETCH_SIZE / WRITE_SIZE / TCC_EA0_ATOMIC_sum / TCC_HIT+MISS) -> gpurun_out/<tag>_mem_<i>/
#        SQ / TCP bottleneck counters                                                 -> gpurun_out/<tag>_sq_<i>/
# Summarise afterwards (here or on the dev box): tools/pmc_summary.py gpurun_out/<tag>_mem profiles/rNN_pmc_hot_path gpurun_out/<tag>_sq
#                                                tools/pmc_table.py   gpurun_out/<tag>_sq  > profiles/rNN_bottleneck_counters.md
# usage: tools/collect_profiles.sh <tag>
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-prof}
mkdir -p $R/gpurun_out
cd $R
timeout -k 10 400 python3 bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err || { echo "bench failed"; exit 1; }
echo "bench done"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_stats -o s -- python3 $R/bench.py --repeats 1 --warmup 0 --cpu-sample 0 --host-path-frames 0 > $R/gpurun_out/${TAG}_stats.json 2> $R/gpurun_out/${TAG}_stats.err || { echo "stats failed"; exit 1; }
echo "stats done"
cd $R
tools/pmc_passes.sh ${TAG}_mem "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_ATOMIC_sum" "TCC_HIT_sum TCC_MISS_sum" || exit 1
tools/pmc_passes.sh ${TAG}_sq "GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
  "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" \
  "SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" \
  "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TOTAL_ACCESSES_sum" \
  "TCP_TCC_ATOMIC_WITH_RET_REQ_sum TCP_TCC_ATOMIC_WITHOUT_RET_REQ_sum TCP_TCC_WRITE_REQ_sum TCC_REQ_sum" \
  "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_ATOMIC_sum" || exit 1
python3 tools/pmc_summary.py gpurun_out/${TAG}_mem gpurun_out/${TAG}_pmc_hot_path gpurun_out/${TAG}_sq > /dev/null
python3 tools/pmc_table.py gpurun_out/${TAG}_sq > gpurun_out/${TAG}_bottleneck_counters.md
echo "profiles done"
