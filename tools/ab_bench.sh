#!/bin/bash
# GPU box: the bench line itself (un-profiled, median of 10 passes) for several builds of libhfpf.so on one box -- for changes whose effect
# sits between kernels (launch counts, host waits) and does not show in tools/ab_kernels.sh's per-kernel averages.
# usage: [AB_ARGS="--workload c3"] tools/ab_bench.sh <name>=<path to libhfpf.so> ...
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out
for spec in "$@"; do
  name=${spec%%=*}; lib=${spec#*=}
  case $lib in /*) ;; *) lib=$R/$lib ;; esac
  HFPF_LIB=$lib timeout -k 10 300 python3 $R/bench.py --cpu-sample 0 --host-path-frames 0 $AB_ARGS > $R/gpurun_out/abb_$name.json 2> $R/gpurun_out/abb_$name.err || { echo "$name failed"; tail -3 $R/gpurun_out/abb_$name.err; continue; }
  python3 - "$name" "$R/gpurun_out/abb_$name.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
print("%s | value %.0f (min %.0f max %.0f) Mpts/s | pass %.3f ms | clean %.3f ms | integrate call %.4f ms | frac %.4f | points_direct %d" % (
    sys.argv[1], d["value"], d["value_min"], d["value_max"], d["ms_per_step"] * d["steps"], d["clean_s"] * 1e3, d["roofline"]["avg_launch_ms"], d["roofline"]["frac"], d["counters"]["points_direct"]))
PY
done
