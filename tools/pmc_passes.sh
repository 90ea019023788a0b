#!/bin/bash
# GPU box: one rocprofv3 --pmc pass per counter set (kernel-trace only beside it) over one timed pass of the default bench.
# usage: [BENCH_ARGS="--workload c3"] tools/pmc_passes.sh <out-prefix> "<set 1>" "<set 2>" ...   (results under gpurun_out/<prefix>_<i>/)
# A pass whose counter set the hardware cannot schedule fails fast and is skipped; a timeout stops the script.
R=${GRAFT_REPO_ROOT:-/root/repo}
P=$1; shift
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
i=0
for SET in "$@"; do
  i=$((i+1))
  timeout -k 10 240 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $R/gpurun_out/${P}_$i -o p -- python3 $R/bench.py --repeats 1 --warmup 0 --cpu-sample 0 --host-path-frames 0 $BENCH_ARGS > $R/gpurun_out/${P}_$i.json 2> $R/gpurun_out/${P}_$i.err
  rc=$?
  echo "pass $i rc=$rc: $SET"
  echo "$SET" > $R/gpurun_out/${P}_$i.set
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "timeout: stopping"; exit 1; fi
done
exit 0
