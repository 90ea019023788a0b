#!/usr/bin/env python3
"""Mean counter value per kernel launch from the passes of tools/pmc_passes.sh.
usage: python tools/pmc_table.py gpurun_out/<prefix> [kernel-substring ...]
Steady-state launches only: of k_integrate the first two dispatches (dry run, first epoch) and the last (100-frame tail) are left
out, of k_buffer the first (first epoch) and the last, of k_update the last; other kernels: all dispatches."""
import collections
import csv
import glob
import sys


def main():
    prefix = sys.argv[1]
    want = sys.argv[2:] or ["k_integrate", "k_update"]
    rows = collections.OrderedDict()
    for f in sorted(glob.glob(prefix + "_*/**/*counter_collection.csv", recursive=True)):
        per = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            for w in want:
                if w in r["Kernel_Name"] and not (w == "k_integrate" and "k_integrate_overflow" in r["Kernel_Name"]):
                    per[(w, r["Counter_Name"])].append(float(r["Counter_Value"]))
        for (w, c), v in per.items():
            lo, hi = {"k_integrate": (2, -1), "k_buffer": (1, -1), "k_update": (0, -1)}.get(w, (0, None))
            vv = v[lo:hi] if len(v) > lo + 2 else v
            rows[(w, c)] = sum(vv) / len(vv)
    print("| kernel | counter | mean per steady-state launch |\n|---|---|---|")
    for (w, c), v in rows.items():
        print("| `%s` | %s | %.4g |" % (w, c, v))


if __name__ == "__main__":
    main()
