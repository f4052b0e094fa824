#!/usr/bin/env python3
"""Table from tools/pmc_fem.sh output directories: per stand-in the kernel's duration (rocprofv3 kernel trace), the fabric requests, moved bytes
(128 B per read request + WRITE_SIZE), their rate, and the wait split.  usage: tools/pmc_fem_table.py gpurun_out/pmc_fem_<tag> [more dirs]"""
import csv, glob, json, os, re, sys

def kernel_row(d):
    best = None
    with open(os.path.join(d, "kernel_stats_spmv.csv")) as f:
        for row in csv.DictReader(f):
            if any(s in row["Name"] for s in ("rowblock_stream", "flat_tile_kernel", "plus_kernel")) and (best is None or int(row["Calls"]) > int(best["Calls"])):
                best = row
    return best

def counters(d, kernel_prefix):
    out = {}
    p = os.path.join(d, "counters.txt")
    if not os.path.exists(p):
        return out
    for ln in open(p):
        m = re.match(r"(\S.*?) (\w+) dispatches=(\d+) mean_last8=(\S+)", ln.strip())
        if m and m.group(1).startswith(kernel_prefix):
            out[m.group(2)] = float(m.group(4))
    return out

print("| run | stand-in | kernel | us (trace) | alg MB | RDREQ 128B (M) | moved MB | moved / alg | TB/s moved | alg frac of 8 TB/s | wait any | wait inst | active | VALU / wave |")
print("|---|---|---|---|---|---|---|---|---|---|---|---|---|---|")
for top in sys.argv[1:]:
    for d in sorted(glob.glob(os.path.join(top, "*"))):
        if not os.path.exists(os.path.join(d, "plain.json")):
            continue
        info = json.load(open(os.path.join(d, "plain.json")))
        k = kernel_row(d)
        if not k:
            continue
        name = re.sub(r"\(.*", "", k["Name"].replace("void ", "").replace("spmv_acc::(anonymous namespace)::", ""))
        c = counters(d, name.split("<")[0])
        us = float(k["AverageNs"]) / 1e3
        alg = info["algorithmic_bytes"] / 1e6
        rd = c.get("TCC_EA0_RDREQ_128B_sum", 0.0) + c.get("TCC_EA0_RDREQ_64B_sum", 0.0) / 2
        moved = rd * 128 / 1e6 + c.get("WRITE_SIZE", 0.0) * 1024 / 1e6
        wc = c.get("SQ_WAVE_CYCLES", 0.0)
        waves = info["nnz"] / 2048 * 4  # (row blocks: about one workgroup of four waves per 1500-2048 non-zeros; a scale for VALU / wave only)
        f = lambda x: f"{x:.3f}"
        print(f"| {os.path.basename(top)} | {info['workload']} | {name} | {us:.2f} | {alg:.1f} | {rd/1e6:.3f} | {moved:.1f} | {f(moved/alg) if moved else '-'} | "
              f"{f(moved/us) if moved else '-'} | {f(alg/us/8)} | {f(c.get('SQ_WAIT_ANY',0)/wc) if wc else '-'} | {f(c.get('SQ_WAIT_INST_ANY',0)/wc) if wc else '-'} | "
              f"{f(c.get('SQ_ACTIVE_INST_ANY',0)/wc) if wc else '-'} | {c.get('SQ_INSTS_VALU',0)/max(waves,1):.0f} |")
