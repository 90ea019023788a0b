#!/bin/bash
# Runs on the GPU box (gpurun): kernel-trace stats of the default bench, then one --pmc pass per counter set
# (never combined with other trace domains), all into gpurun_out/.  Summarise afterwards with tools/pmc_summary.py.
set -e -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/stats3 -o s3 -- python3 $R/bench.py --cpu-sample 0 --host-path-frames 0 > $R/gpurun_out/stats3.json 2> $R/gpurun_out/stats3.err
echo "stats done"
for C in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_ATOMIC_sum" "TCC_HIT_sum TCC_MISS_sum"; do
  N=$(echo $C | tr ' ' '_')
  timeout -k 10 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/gpurun_out/pmc3_$N -o p -- python3 $R/bench.py --steps 600 --warmup 5 --cpu-sample 0 --host-path-frames 0 > $R/gpurun_out/pmc3_$N.json 2> $R/gpurun_out/pmc3_$N.err
  echo "pmc $N done"
done
