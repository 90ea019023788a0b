#!/usr/bin/env python3
"""Dev box: rewrite the generated parts of profiles/README.md's round-3 section (kernel table, source hash) from the committed files."""
import csv, json, os, re
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
p = os.path.join(R, "profiles", "README.md")
s = open(p).read()
sha = json.load(open(os.path.join(R, "profiles", "r03_pmc_hot_path.json")))["source_sha"]
s = re.sub(r"(## Round 3 \(final build of the round: kernel sources `)[0-9a-f]{16}(`)", r"\g<1>%s\g<2>" % sha, s)
rows = list(csv.DictReader(open(os.path.join(R, "profiles", "r03_kernel_stats.csv"))))
tab = []
for r in rows[:17]:
    n = r["Name"].split("(")[0].replace("void ", "")
    if n == "":
        n = "k_publish_counters (counter mailbox)"
    if n.startswith("rocprim"):
        n = "rocprim radix-sort / scan kernel"
    tab.append("| `%s` | %s | %.3f | %.1f | %.1f |" % (n[:64], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
head = "### Kernel totals of two 1000-frame passes (`r03_kernel_stats.csv`)\n\n| kernel | calls | total ms | avg µs | % |\n|---|---|---|---|---|\n"
a = s.index(head) + len(head)
b = s.index("\n\n`__amd_rocclr_fillBufferAligned` is dominated", a)
s = s[:a] + "\n".join(tab) + s[b:]
open(p, "w").write(s)
print("\n".join(tab[:8]))
