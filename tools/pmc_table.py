#!/usr/bin/env python3
"""Mean counter value per kernel launch from the passes of tools/pmc_passes.sh.
usage: python tools/pmc_table.py gpurun_out/<prefix> [kernel-substring ...]   (steady-state launches: the first one of each kernel is skipped
when the kernel ran more than twice)"""
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
                if w in r["Kernel_Name"]:
                    per[(w, r["Counter_Name"])].append(float(r["Counter_Value"]))
        for (w, c), v in per.items():
            vv = v[1:] if len(v) > 2 else v
            rows[(w, c)] = sum(vv) / len(vv)
    print("| kernel | counter | mean per steady-state launch |\n|---|---|---|")
    for (w, c), v in rows.items():
        print("| `%s` | %s | %.4g |" % (w, c, v))


if __name__ == "__main__":
    main()
