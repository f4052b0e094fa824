#!/usr/bin/env python3
"""The memory skeleton of one kernel instance as hipcc compiled it: every load / store / barrier and every s_waitcnt, in program order.
Round 6's largest single gain came from reading these lines (a wave-uniform branch between two loads makes the compiler's wait-count pass give up the
exact count: `s_waitcnt vmcnt(0)` where `vmcnt(5)` would do -- profiles/r06_col16_counters.md section 5); tests/test_host_logic.py pins the result.
usage: tools/wait_counts.py <file.hip under spmv_acc_amd/csrc> '<demangled instance prefix>' [extra hipcc flags ...]
   e.g. tools/wait_counts.py k_rowblock.hip 'rowblock_stream_kernel<4, true, true, false, false, true>'"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, want = sys.argv[1], sys.argv[2]
csrc = os.path.join(ROOT, "spmv_acc_amd", "csrc")
with tempfile.TemporaryDirectory() as tmp:
    out = os.path.join(tmp, "k.s")
    r = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-DKERNEL_STRATEGY_ADAPTIVE", "-I" + os.path.join(ROOT, "include"),
                        "--cuda-device-only", "-S", os.path.join(csrc, src), "-o", out] + sys.argv[3:], capture_output=True, text=True)
    if r.returncode != 0:
        sys.exit(r.stderr[-3000:])
    asm = open(out).read()
labels = re.findall(r"^(_Z\w+):", asm, flags=re.M)
names = subprocess.run(["c++filt"], input="\n".join(labels), capture_output=True, text=True).stdout.split("\n")
hits = [(l, n) for l, n in zip(labels, names) if want in n.replace("spmv_acc::(anonymous namespace)::", "")]
if not hits:
    sys.exit("no instance matches; candidates:\n" + "\n".join(sorted(set(n.split("(")[0] for n in names if "kernel" in n))[:80]))
label, name = hits[0]
body = asm[asm.index("\n" + label + ":"):]
body = body[:body.index("s_endpgm")]
print("#", name.split("(")[0])
keep = re.compile(r"s_waitcnt|global_load|global_store|global_atomic|buffer_load|buffer_store|s_load_|s_barrier|ds_read|ds_write|ds_bpermute|s_cbranch|^\.LBB")
for i, ln in enumerate(body.split("\n")):
    t = ln.strip()
    if keep.search(t):
        print(f"{i:5d}  {t.split(';')[0].rstrip()}")
