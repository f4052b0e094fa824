#!/usr/bin/env python3
"""Per-kernel register table from hipcc's -Rpass-analysis=kernel-resource-usage remarks (read from a file or stdin).
usage: hipcc ... -Rpass-analysis=kernel-resource-usage -c k_rowblock.hip 2> usage.txt; tools/resource_table.py usage.txt [filter]
Prints: demangled instance, VGPRs, AGPRs, SGPRs, scratch bytes, waves/SIMD.  tests/test_host_logic.py uses parse() to pin the settled
instances at <= 64 VGPRs (8 waves per SIMD)."""
import re
import subprocess
import sys


def parse(text):
    out = []
    for rec in re.split(r"remark: Function Name: ", text)[1:]:
        mangled = rec.split()[0]
        g = lambda k: int(re.search(k + r": (\d+)", rec).group(1))
        out.append({"mangled": mangled, "vgprs": g("VGPRs"), "agprs": g("AGPRs"), "sgprs": g("SGPRs"),
                    "scratch": g(r"ScratchSize \[bytes/lane\]"), "occupancy": g(r"Occupancy \[waves/SIMD\]")})
    names = subprocess.run(["c++filt"], input="\n".join(r["mangled"] for r in out), capture_output=True, text=True).stdout.split("\n")
    for r, n in zip(out, names):
        r["name"] = n.replace("spmv_acc::(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    return out


if __name__ == "__main__":
    text = open(sys.argv[1]).read() if len(sys.argv) > 1 and sys.argv[1] != "-" else sys.stdin.read()
    pat = sys.argv[2] if len(sys.argv) > 2 else ""
    for r in parse(text):
        if pat in r["name"]:
            print(f"{r['name']:70s} V {r['vgprs']:3d} A {r['agprs']:2d} S {r['sgprs']:3d} scratch {r['scratch']:3d} waves/SIMD {r['occupancy']}")
