#!/bin/bash
# GPU box: per-kernel average durations (rocprofv3 --kernel-trace --stats) of the default bench for several builds of libhfpf.so,
# for A/B decisions below the bench's ~1.5 % run-to-run noise.
# usage: [AB_ARGS="--workload c3"] tools/ab_kernels.sh <name>=<path to libhfpf.so> ...   (results: gpurun_out/ab_<name>/, one summary line per build)
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for spec in "$@"; do
  name=${spec%%=*}; lib=${spec#*=}
  case $lib in /*) ;; *) lib=$R/$lib ;; esac   # (the profiler runs from /tmp)
  rm -rf $R/gpurun_out/ab_$name
  HFPF_LIB=$lib timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/ab_$name -o s -- python3 $R/bench.py --repeats 3 --warmup 0 --cpu-sample 0 --host-path-frames 0 $AB_ARGS > $R/gpurun_out/ab_$name.json 2> $R/gpurun_out/ab_$name.err || { echo "$name failed"; tail -3 $R/gpurun_out/ab_$name.err; continue; }
  python3 - "$name" "$R/gpurun_out/ab_$name" <<'PY'
import csv, glob, sys
name, d = sys.argv[1], sys.argv[2]
f = glob.glob(d + "/**/*kernel_stats.csv", recursive=True)[0]
out = []
for r in csv.DictReader(open(f)):
    n = r["Name"]
    for k in ("k_integrate_overflow", "k_integrate<", "k_update", "k_replay", "k_register", "k_buffer", "k_normal", "k_depinc_offsets", "k_depinc_fill", "k_clean_begin", "k_gate"):
        if k in n:
            k = k.rstrip("<")
            out.append("%s %.1f us x%s" % (k, float(r["AverageNs"]) / 1e3, r["Calls"]))
print(name, "|", " | ".join(out))
PY
done
