#!/usr/bin/env python3
"""Largest engine-vs-oracle deviation per float column over the comparisons a test run logged with HFPF_ERR_REPORT=<file>
(tests/scenes.py error_report): the tolerance ledger of DESIGN.md section 5.  usage: tools/err_report_max.py <file> ..."""
import json
import sys

for path in sys.argv[1:]:
    mx, n = {}, 0
    for line in open(path):
        r = json.loads(line)
        n += 1
        for k, v in r.items():
            if k != "rows":
                mx[k] = max(mx.get(k, 0.0), v)
    print(path, "comparisons", n)
    for k in sorted(mx):
        print("  %-14s %.3g" % (k, mx[k]))
