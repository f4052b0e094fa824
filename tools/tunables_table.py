#!/usr/bin/env python3
"""INTEGRATION.md section 5's table, regenerated from the tunable table of spmv_acc_amd/csrc/config.cpp (name, default, the entry's comment).
usage: tools/tunables_table.py            print the table
       tools/tunables_table.py --write    replace the table in INTEGRATION.md in place"""
import os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = open(os.path.join(ROOT, "spmv_acc_amd", "csrc", "config.cpp")).read()
table = src[src.index("Tunable g_tunables[] = {") + len("Tunable g_tunables[] = {"):src.index("static_assert(sizeof(g_tunables)")]
rows, cur = [], None
for ln in table.splitlines():
    m = re.match(r'\s*\{"(\w+)",\s*([^,]+),\s*[^}]+\},?\s*(?://\s*(.*))?$', ln)
    if m:
        cur = [m.group(1), m.group(2).strip(), (m.group(3) or "").strip()]
        rows.append(cur)
    elif cur is not None and ln.strip().startswith("//"):
        cur[2] += " " + ln.strip()[2:].strip()
consts = {"kFlatReduceBuilt": "0 (1 when built with FLAT_SEGMENT_SUM_REDUCE)", "kMaxGridBlocks": "8388593", "kFlatSmallNnz >> 10": "24576"}
out = ["| name | default | meaning |", "|---|---|---|"]
for name, default, text in rows:
    out.append(f"| `{name}` | {consts.get(default, default)} | {text.replace('|', '/')} |")
text = "\n".join(out)
if "--write" in sys.argv:
    p = os.path.join(ROOT, "INTEGRATION.md")
    doc = open(p).read()
    i = doc.index("| name | default | meaning |")
    j = i
    lines = doc[i:].split("\n")
    k = 0
    while k < len(lines) and lines[k].startswith("|"):
        k += 1
    doc = doc[:i] + text + "\n" + "\n".join(lines[k:])
    open(p, "w").write(doc)
    print(f"{len(rows)} tunables written to INTEGRATION.md")
else:
    print(text)
