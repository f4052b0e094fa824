"""Regenerates the committed golden fixtures.  Run in the build container (needs oracle/liboracle.so and,
for the analysis goldens, oracle/_ref/libref_analyze.so built from /root/reference):

    python tests/golden/make_golden.py

  spmv_cases.npz      inputs (CSR, x, y0) + expected y for several (alpha, beta), produced by our CPU
                      restatement of cli/verification.cpp:56-66 and cross-checked here against scipy.
                      (The reference ships no golden vectors and its host_spmv TU needs a CMake-generated
                      header, so these are NOT reference-run outputs: "parity unpinned", see DESIGN.md.)
  analysis_cases.npz  rowptr inputs + break_points / first_block_of_row / block count produced by the
                      REFERENCE's own csr_adaptive_plus_analyze.cpp (compiled unmodified into oracle/_ref).
  breakpoint_cases.npz rowptr inputs + break points from our restatement of flat_imp.inl:108-131.
Fixtures are data only (inputs and expected outputs).
"""
import os
import sys

import numpy as np
import scipy.sparse as sp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import oracle_lib  # noqa: E402
from spmv_acc_amd import synth  # noqa: E402

ALPHA_BETA = [(1.0, 1.0), (0.5, -2.0), (1.0, 0.0), (-1.25, 0.75), (0.0, 1.0)]


def spmv_cases():
    out = {}
    specs = [
        ("uniform5", dict(m=600, n=640, avg=5, kind="uniform")),
        ("short", dict(m=900, n=900, avg=2, kind="short")),
        ("powerlaw", dict(m=500, n=700, avg=6, kind="powerlaw")),
        ("spikes", dict(m=400, n=3000, avg=3, kind="spikes")),
        ("empty_rows", dict(m=800, n=500, avg=4, kind="empty_rows")),
        ("dense_rows", dict(m=12, n=2000, avg=400, kind="dense_rows")),
    ]
    for i, (name, kw) in enumerate(specs):
        rowptr, cols, vals = synth.random_csr(seed=1000 + i, **kw)
        rng = np.random.default_rng(2000 + i)
        n = kw["n"]
        m = kw["m"]
        x = synth.reference_rand_grid(n, rng)
        y0 = synth.reference_rand_grid(m, rng)
        out[f"{name}__rowptr"] = rowptr
        out[f"{name}__cols"] = cols
        out[f"{name}__vals"] = vals
        out[f"{name}__x"] = x
        out[f"{name}__y0"] = y0
        A = sp.csr_matrix((vals, cols, rowptr), shape=(m, n))
        for k, (a, b) in enumerate(ALPHA_BETA):
            y = oracle_lib.host_spmv(a, b, rowptr, cols, vals, x, y0)
            ys = a * (A @ x) + b * y0  # independent implementation, different summation order
            err = oracle_lib.scaled_error(y, ys, a, b, rowptr, cols, vals, x, y0)
            assert err < 1e-13, (name, a, b, err)
            out[f"{name}__out{k}"] = y
    out["alpha_beta"] = np.array(ALPHA_BETA)
    out["names"] = np.array([s[0] for s in specs])
    return out


def rowptr_cases():
    rng = np.random.default_rng(31337)
    cases = {}

    def add(name, lens):
        rp = np.zeros(len(lens) + 1, dtype=np.int32)
        np.cumsum(lens, out=rp[1:])
        cases[name] = rp

    add("short", rng.integers(0, 12, 3000))
    add("pareto", np.minimum((rng.pareto(1.2, 2500) * 3).astype(np.int64), 20000))
    lens = rng.integers(0, 3, 2000)
    lens[rng.integers(0, 2000, 4)] = rng.integers(4096, 30000, 4)
    add("few_long", lens)
    add("mid", rng.integers(0, 700, 600))
    lens = np.zeros(1500, dtype=np.int64)
    lens[700] = 9000  # long row preceded by empty rows in the same block
    lens[701] = 4096
    lens[1400:] = 5
    add("leading_empty_long", lens)
    add("exact_multiples", np.full(64, 1024, dtype=np.int64))
    add("one_row", np.array([10000]))
    add("all_empty", np.zeros(300, dtype=np.int64))
    return cases


def analysis_cases():
    assert oracle_lib.ref() is not None, "oracle/_ref not built: run `make -C oracle` with /root/reference present"
    out = {}
    params = [(512, 1, 2048), (512, 2, 2048), (512, 8, 2048), (512, 64, 2048), (256, 4, 1024), (1024, 16, 4096)]
    cases = rowptr_cases()
    out["names"] = np.array(list(cases))
    out["params"] = np.array(params, dtype=np.int32)
    for name, rp in cases.items():
        out[f"{name}__rowptr"] = rp
        for k, (threads, vec, min_nnz) in enumerate(params):
            blocks, bp, fbr = oracle_lib.ref_adaptive_plus_analyze(rp, min_nnz, threads, vec)
            # our restatement must agree with the reference before the fixture is written
            b2, bp2, fbr2 = oracle_lib.adaptive_plus_analyze(rp, min_nnz, threads, vec)
            assert blocks == b2 and np.array_equal(bp, bp2) and np.array_equal(fbr, fbr2), (name, threads, vec)
            out[f"{name}__{k}__bp"] = bp
            out[f"{name}__{k}__fbr"] = fbr
    return out


def breakpoint_cases():
    out = {}
    cases = rowptr_cases()
    strides = [1024, 2048, 256]
    out["names"] = np.array(list(cases))
    out["strides"] = np.array(strides, dtype=np.int32)
    for name, rp in cases.items():
        out[f"{name}__rowptr"] = rp
        for s in strides:
            out[f"{name}__{s}"] = oracle_lib.break_points(rp, s)
    return out


if __name__ == "__main__":
    np.savez_compressed(os.path.join(HERE, "spmv_cases.npz"), **spmv_cases())
    np.savez_compressed(os.path.join(HERE, "analysis_cases.npz"), **analysis_cases())
    np.savez_compressed(os.path.join(HERE, "breakpoint_cases.npz"), **breakpoint_cases())
    for f in ("spmv_cases.npz", "analysis_cases.npz", "breakpoint_cases.npz"):
        print(f, os.path.getsize(os.path.join(HERE, f)), "bytes")


# ---- reader fixtures: input FILES + what the REFERENCE's own readers parse from them --------------------------------------
def reader_case_files(tmpdir):
    """Small matrix files in the three formats of cli/main.cpp:36-40 (written here, not taken from the reference)."""
    import struct

    files = {}
    rowptr, cols, vals = synth.random_csr(60, 50, 4, seed=77, kind="powerlaw")
    x = synth.reference_rand_grid(50, np.random.default_rng(5))
    p = os.path.join(tmpdir, "small.csr")
    with open(p, "w") as f:
        f.write("% any header text 60 50\n")
        f.write(" ".join(repr(float(t)) for t in vals) + "\n")
        f.write(" ".join(str(int(t)) for t in cols) + "\n")
        f.write(" ".join(str(int(t)) for t in rowptr) + "\n")
        f.write(" ".join(repr(float(t)) for t in x) + "\n")
    files["small.csr"] = "csr"
    for name, valtype in (("real.bin2", 3), ("pattern.bin2", 1)):
        p = os.path.join(tmpdir, name)
        with open(p, "wb") as f:
            f.write(struct.pack("<6i", 0x20211015, 2, valtype, 60, 50, len(cols)))
            rowptr.astype("<i4").tofile(f)
            cols.astype("<i4").tofile(f)
            if valtype == 3:
                vals.astype("<f8").tofile(f)
        files[name] = "bin2"
    rng = np.random.default_rng(9)
    ent = sorted({(int(r), int(c)) for r, c in zip(rng.integers(1, 41, 150), rng.integers(1, 31, 150))})  # no duplicates
    with open(os.path.join(tmpdir, "general.mtx"), "w") as f:
        f.write("%%%%MatrixMarket matrix coordinate real general\n%% comment line\n40 30 %d\n" % len(ent))
        order = rng.permutation(len(ent))
        for k in order:
            f.write("%d %d %.17g\n" % (ent[k][0], ent[k][1], rng.standard_normal()))
    files["general.mtx"] = "mtx"
    low = sorted({(max(r, c), min(r, c)) for r, c in zip(rng.integers(1, 36, 120), rng.integers(1, 36, 120))})
    with open(os.path.join(tmpdir, "symmetric.mtx"), "w") as f:
        f.write("%%%%MatrixMarket matrix coordinate real symmetric\n35 35 %d\n" % len(low))
        for r, c in low:
            f.write("%d %d %.17g\n" % (r, c, rng.standard_normal()))
    files["symmetric.mtx"] = "mtx"
    with open(os.path.join(tmpdir, "pattern.mtx"), "w") as f:
        f.write("%%%%MatrixMarket matrix coordinate pattern general\n40 30 %d\n" % len(ent))
        for r, c in ent:
            f.write("%d %d\n" % (r, c))
    files["pattern.mtx"] = "mtx"
    with open(os.path.join(tmpdir, "integer.mtx"), "w") as f:
        f.write("%%%%MatrixMarket matrix coordinate integer general\n40 30 %d\n" % len(ent))
        for k, (r, c) in enumerate(ent):
            f.write("%d %d %d\n" % (r, c, (k % 9) - 4))
    files["integer.mtx"] = "mtx"
    return files


def reader_cases():
    import tempfile

    assert oracle_lib.ref_readers() is not None, "oracle/_ref/libref_readers.so not built"
    out = {}
    with tempfile.TemporaryDirectory() as d:
        files = reader_case_files(d)
        out["names"] = np.array(list(files))
        out["formats"] = np.array([files[n] for n in files])
        for name, fmt in files.items():
            path = os.path.join(d, name)
            out[f"{name}__file"] = np.frombuffer(open(path, "rb").read(), dtype=np.uint8)
            rows, cols, nnz, rp, ci, v, x = oracle_lib.ref_read_matrix(path, fmt)
            out[f"{name}__dims"] = np.array([rows, cols, nnz], dtype=np.int64)
            out[f"{name}__rowptr"], out[f"{name}__colidx"], out[f"{name}__values"], out[f"{name}__x"] = rp, ci, v, x
    return out


if __name__ == "__main__":
    np.savez_compressed(os.path.join(HERE, "reader_cases.npz"), **reader_cases())
    print("reader_cases.npz", os.path.getsize(os.path.join(HERE, "reader_cases.npz")), "bytes")
