#!/usr/bin/env python3
"""Tabulate gpurun_out/bn_*/p_counter_collection.csv (tools/collect_bottleneck.sh): per counter, the mean over the
steady-state dispatches of k_integrate and k_update (600-frame bench, 150 frames per launch: dispatch 0 is the warm-up,
1 the buffering epoch, 2.. steady state).  usage: python tools/bottleneck_summary.py gpurun_out [out.md]"""
import collections
import csv
import glob
import sys


def main():
    root = sys.argv[1]
    vals = collections.OrderedDict()
    durs = {}
    for d in sorted(glob.glob(root + "/bn_*/")):
        per = collections.defaultdict(lambda: collections.defaultdict(dict))
        for r in csv.DictReader(open(d + "p_counter_collection.csv")):
            k = r["Kernel_Name"]
            which = "k_integrate" if "k_integrate" in k else ("k_update" if "k_update" in k else None)
            if which:
                per[which][r["Counter_Name"]][int(r["Dispatch_Id"])] = float(r["Counter_Value"])
        for which, ctrs in per.items():
            for c, byd in ctrs.items():
                ids = sorted(byd)
                steady = ids[2:] if which == "k_integrate" else ids  # k_update only exists in steady state
                if steady:
                    vals[(which, c)] = sum(byd[i] for i in steady) / len(steady)
        kt = collections.defaultdict(list)
        for r in csv.DictReader(open(d + "p_kernel_trace.csv")):
            k = r["Kernel_Name"]
            which = "k_integrate" if "k_integrate" in k else ("k_update" if "k_update" in k else None)
            if which:
                kt[which].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
        for which, v in kt.items():
            v.sort()
            steady = v[2:] if which == "k_integrate" else v
            durs.setdefault(which, []).append(sum(x[1] for x in steady) / max(len(steady), 1) / 1e6)
    lines = ["| kernel | counter | mean per steady-state launch |", "|---|---|---|"]
    for (which, c), v in vals.items():
        lines.append("| `%s` | %s | %.4g |" % (which, c, v))
    for which, v in durs.items():
        lines.append("| `%s` | duration under --pmc (ms, min..max over passes) | %.3f .. %.3f |" % (which, min(v), max(v)))
    out = "\n".join(lines)
    print(out)
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(out + "\n")


if __name__ == "__main__":
    main()
