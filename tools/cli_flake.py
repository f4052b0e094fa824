import subprocess, sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
from spmv_acc_amd import synth
from test_cli_io import write_bin2
rowptr, cols, vals = synth.random_csr(60000, 60000, 9, seed=8, kind="powerlaw")
p = "/tmp/b.bin2"; write_bin2(p, 60000, 60000, rowptr, cols, vals)
cli = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "spmv_acc_amd", "bin", "spmv-cli")
bad = 0
for i in range(25):
    r = subprocess.run([cli, p, "-f", "bin2", "--benchmark"], capture_output=True, text=True)
    rows = [l.split(",") for l in r.stdout.splitlines() if l.startswith("PERFORMANCE,")][1:]
    plans = [l for l in r.stdout.splitlines() if l.startswith("PLAN,")]
    fails = [x for x in rows if int(x[-2]) != 0 or float(x[11]) != 0.0]
    if r.returncode != 0 or fails or len(plans) != len(rows):
        bad += 1
        print("RUN", i, "rc", r.returncode, "fails", fails, "nplans", len(plans), "nrows", len(rows), r.stderr[-300:], flush=True)
        for l in plans: print("   ", l)
print("bad runs:", bad)
