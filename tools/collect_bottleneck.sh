#!/bin/bash
# GPU box: one rocprofv3 --pmc pass per counter set (kernel-trace only beside it) over a 600-frame default bench.
# A pass whose counter set the hardware cannot schedule fails fast and is skipped; a timeout stops the script.
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
i=0
while read -r SET; do
  [ -z "$SET" ] && continue
  i=$((i+1))
  timeout -k 10 240 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $R/gpurun_out/bn_$i -o p -- python3 $R/bench.py --steps 600 --warmup 5 --cpu-sample 0 --host-path-frames 0 > $R/gpurun_out/bn_$i.json 2> $R/gpurun_out/bn_$i.err
  rc=$?
  echo "pass $i rc=$rc: $SET"
  echo "$SET" > $R/gpurun_out/bn_$i.set
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "timeout: stopping"; exit 1; fi
done <<'SETS'
GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM
SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_ACTIVE_INST_LDS
TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum
TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_WAVEFRONTS_sum
TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TOTAL_ACCESSES_sum
TCP_TCC_ATOMIC_WITH_RET_REQ_sum TCP_TCC_ATOMIC_WITHOUT_RET_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_WRITE_REQ_LATENCY_sum
TCC_REQ_sum TCC_BUSY_avr TCC_TAG_STALL_sum TCC_ATOMIC_sum
TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_WRREQ_sum
SETS
