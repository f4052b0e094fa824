"""CLI + matrix readers (SURVEY.md 8f 'next', rank 2-3): .csr text, bin2, MatrixMarket through spmv-cli.
CPU tests use --print-stats / --no-gpu (BASELINE.json configs[0]: the CPU-side verification path);
the GPU tests run the reference's CLI and benchmark protocols on the device."""
import os
import struct
import subprocess

import numpy as np
import pytest

from spmv_acc_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "spmv_acc_amd", "bin", "spmv-cli")


@pytest.fixture(scope="module")
def cli(hiplib):
    if not os.path.exists(CLI):
        import __graft_entry__

        __graft_entry__.build()
    return CLI


def write_csr_text(path, rowptr, cols, vals, x, header="synthetic"):
    with open(path, "w") as f:
        f.write(header + "\n")
        f.write(" ".join(repr(float(t)) for t in vals) + "\n")
        f.write(" ".join(str(int(t)) for t in cols) + "\n")
        f.write(" ".join(str(int(t)) for t in rowptr) + "\n")
        f.write(" ".join(repr(float(t)) for t in x) + "\n")


def write_bin2(path, rows, cols_n, rowptr, cols, vals, valtype=3):
    with open(path, "wb") as f:
        f.write(struct.pack("<6i", 0x20211015, 2, valtype, rows, cols_n, len(cols)))
        np.asarray(rowptr, dtype="<i4").tofile(f)
        np.asarray(cols, dtype="<i4").tofile(f)
        if valtype == 2:
            np.asarray(vals, dtype="<i4").tofile(f)
        elif valtype in (3, 4):
            np.asarray(vals, dtype="<f8").tofile(f)


def stats(cli, path, fmt):
    r = subprocess.run([cli, path, "-f", fmt, "--print-stats"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    kv = dict(t.split("=") for t in r.stdout.split()[1:])
    return {k: float(v) for k, v in kv.items()}


def test_csr_text_reader(cli, tmp_path):
    rowptr, cols, vals = synth.random_csr(300, 280, 6, seed=3, kind="powerlaw")
    x = np.random.default_rng(1).standard_normal(280)
    p = str(tmp_path / "a.csr")
    write_csr_text(p, rowptr, cols, vals, x, header="any free-form text 1 2 3")
    s = stats(cli, p, "csr")
    assert (s["rows"], s["cols"], s["nnz"], s["x_len"]) == (300, 280, len(vals), 280)
    assert s["sum_colidx"] == cols.astype(np.int64).sum() and s["sum_rowptr"] == rowptr.astype(np.int64).sum()
    assert abs(s["sum_values"] - float(np.sum(vals.astype(np.longdouble)))) < 1e-9
    assert abs(s["sum_x"] - float(np.sum(x.astype(np.longdouble)))) < 1e-9


def test_bin2_reader_all_value_types(cli, tmp_path):
    rowptr, cols, vals = synth.random_csr(200, 250, 5, seed=4)
    for valtype, v, want in ((3, vals, vals.sum()), (2, np.arange(len(cols)) % 7 - 3, float((np.arange(len(cols)) % 7 - 3).sum())),
                             (1, None, float(len(cols)))):
        p = str(tmp_path / f"a{valtype}.bin2")
        write_bin2(p, 200, 250, rowptr, cols, v, valtype)
        s = stats(cli, p, "bin2")
        assert (s["rows"], s["cols"], s["nnz"]) == (200, 250, len(cols))
        assert s["sum_colidx"] == cols.astype(np.int64).sum()
        assert abs(s["sum_values"] - want) < 1e-9, valtype
    bad = str(tmp_path / "bad.bin2")
    with open(bad, "wb") as f:
        f.write(struct.pack("<6i", 0x12345678, 2, 3, 1, 1, 0))
    r = subprocess.run([cli, bad, "-f", "bin2", "--print-stats"], capture_output=True, text=True)
    assert r.returncode == 3 and "magic" in r.stderr  # fails loudly (the reference returns an empty matrix)


def test_matrix_market_reader(cli, tmp_path):
    p = str(tmp_path / "g.mtx")
    with open(p, "w") as f:
        f.write("%%MatrixMarket matrix coordinate real general\n% comment\n3 4 5\n3 1 1.5\n1 2 -2.0\n1 1 4.0\n2 4 0.25\n3 3 7\n")
    s = stats(cli, p, "mtx")
    assert (s["rows"], s["cols"], s["nnz"]) == (3, 4, 5)
    assert s["sum_rowptr"] == 0 + 2 + 3 + 5 and s["sum_colidx"] == 0 + 1 + 3 + 0 + 2 and s["sum_values"] == 10.75
    p = str(tmp_path / "s.mtx")
    with open(p, "w") as f:
        f.write("%%MatrixMarket matrix coordinate real symmetric\n3 3 4\n1 1 2\n2 1 3\n3 1 -1\n3 3 5\n")
    s = stats(cli, p, "mtx")  # off-diagonals mirrored: 4 + 2 entries
    assert (s["nnz"], s["sum_values"]) == (6, 2 + 3 + 3 - 1 - 1 + 5)
    p = str(tmp_path / "p.mtx")
    with open(p, "w") as f:
        f.write("%%MatrixMarket matrix coordinate pattern general\n2 2 3\n1 1\n1 2\n2 2\n")
    s = stats(cli, p, "mtx")
    assert (s["nnz"], s["sum_values"]) == (3, 3.0)
    r = subprocess.run([cli, p, "-f", "mtx", "--no-gpu"], capture_output=True, text=True)
    assert r.returncode == 0 and "pass 2 validation" in r.stdout


def test_config0_rajat03_like_cpu_verification_path(cli, tmp_path):
    """BASELINE.json configs[0]: a rajat03-sized .csr file through spmv-cli's CPU-side verification path."""
    rowptr, cols, vals = synth.rajat03_like()
    x = synth.reference_rand_grid(7602, np.random.default_rng(0xC1))
    p = str(tmp_path / "rajat03_like.csr")
    write_csr_text(p, rowptr, cols, vals, x, header="rajat03-like 7602 7602")
    r = subprocess.run([cli, p, "-f", "csr", "--no-gpu"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert "Congratulation, pass 7602 validation!" in r.stdout


def test_cli_usage_errors(cli):
    assert subprocess.run([cli], capture_output=True).returncode == 2
    assert subprocess.run([cli, "/nonexistent.csr", "-f", "csr", "--no-gpu"], capture_output=True).returncode == 3
    assert subprocess.run([cli, "x", "-f", "coo"], capture_output=True).returncode == 2


@pytest.mark.gpu
def test_cli_on_gpu_all_formats(cli, tmp_path):
    """cli/main.cpp's protocol on the device: 10 warm-ups, timed run, verify against host_spmv."""
    rowptr, cols, vals = synth.rajat03_like()
    x = synth.reference_rand_grid(7602, np.random.default_rng(0xC1))
    pc = str(tmp_path / "r.csr")
    write_csr_text(pc, rowptr, cols, vals, x)
    pb = str(tmp_path / "r.bin2")
    write_bin2(pb, 7602, 7602, rowptr, cols, vals)
    for path, fmt in ((pc, "csr"), (pb, "bin2")):
        for strat in ("adaptive", "flat", "line_enhance", "default"):
            r = subprocess.run([cli, path, "-f", fmt, "--strategy", strat], capture_output=True, text=True)
            assert r.returncode == 0, (fmt, strat, r.stdout, r.stderr)
            assert "Congratulation, pass 7602 validation!" in r.stdout and "elapsed time:" in r.stdout


@pytest.mark.gpu
def test_benchmark_mode_csv(cli, tmp_path):
    rowptr, cols, vals = synth.random_csr(60000, 60000, 9, seed=8, kind="powerlaw")
    p = str(tmp_path / "b.bin2")
    write_bin2(p, 60000, 60000, rowptr, cols, vals)
    r = subprocess.run([cli, p, "-f", "bin2", "--benchmark"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    lines = [l for l in r.stdout.splitlines() if l.startswith("PERFORMANCE,")]
    header = lines[0].split(",")
    assert header[:7] == ["PERFORMANCE", "matrix name", "strategy name", "rows", "cols", "nnz", "nnz/row"]
    assert header[-3:] == ["first_failed_at", "failed_count", "max_error"] and len(header) == 19
    rows = [l.split(",") for l in lines[1:]]
    assert {r_[2] for r_ in rows} >= {"default", "adaptive", "flat", "line_enhance"}
    for r_ in rows:
        assert len(r_) == 19 and int(r_[3]) == 60000 and int(r_[5]) == len(cols)
        assert int(r_[-2]) == 0 and int(r_[-3]) == -1 and float(r_[12]) > 0  # verified, timed
    # the one-time preparation is on record next to every steady-state line: the first timed run rebuilt the plan, its `pre`
    # is what the library reports for that call (spmv_acc_last_prepare_us), the median-by-total rule then prints a steady run
    plans = [l.split(",") for l in r.stdout.splitlines() if l.startswith("PLAN,")]
    assert {p_[2] for p_ in plans} == {r_[2] for r_ in rows}
    for p_ in plans:
        assert p_[3] == "first_call_us" and p_[5] == "pre_us" and p_[7] == "calc_us"
        assert float(p_[6]) > 0 and float(p_[4]) >= float(p_[6])  # preparation happened and is part of the first call
    for r_ in rows:  # the printed run is the median by total: a steady run (pre 0) unless preparation is below the timing noise
        assert float(r_[11]) == 0.0 or float(r_[11]) < 0.05 * float(r_[15])


@pytest.mark.gpu
def test_cli_device_side_verify(cli, tmp_path):
    """The reference's -DDEVICE_SIDE_VERIFY_FLAG=ON build (config.cmake:9, cli/verification.cpp:81-112): the expected y comes
    from rocSPARSE on the device instead of host_spmv.  Here a run-time switch; rocSPARSE is the checker only (dlopen'ed by
    the CLI, never linked into the library)."""
    # Which rocSPARSE: /opt/rocm's takes ~190 s to load its code objects in a fresh process on a fresh box (measured, round 2:
    # that is why this test used to hide behind SPMV_ACC_SLOW_TESTS); the build bundled with the PyTorch wheel loads in seconds
    # and is the one tests/test_gpu_parity.py::test_against_rocsparse_as_second_opinion already uses in-process.  The CLI takes
    # the path from SPMV_CLI_ROCSPARSE.  (SPMV_ACC_SLOW_TESTS=1 still exercises /opt/rocm's.)
    import torch

    bundled = os.path.join(os.path.dirname(torch.__file__), "lib", "librocsparse.so")
    env = dict(os.environ)
    if os.environ.get("SPMV_ACC_SLOW_TESTS", "0") != "1":
        if not os.path.exists(bundled):
            pytest.skip("no rocSPARSE bundled with this PyTorch; set SPMV_ACC_SLOW_TESTS=1 for /opt/rocm's (minutes)")
        env["SPMV_CLI_ROCSPARSE"] = bundled
    elif not any(os.path.exists(p) for p in ("/opt/rocm/lib/librocsparse.so", "/opt/rocm/lib/librocsparse.so.1")):
        pytest.skip("no rocSPARSE on this box")
    rowptr, cols, vals = synth.rajat03_like()
    x = synth.reference_rand_grid(7602, np.random.default_rng(0xC1))
    pc = str(tmp_path / "r.csr")
    write_csr_text(pc, rowptr, cols, vals, x)
    # (one process: loading rocSPARSE's code objects takes far longer than the SpMVs)
    r = subprocess.run([cli, pc, "-f", "csr", "--benchmark", "--strategy", "flat", "--device-verify"], capture_output=True, text=True,
                       env=env, timeout=420)
    assert r.returncode == 0, (r.stdout, r.stderr)
    row = [l.split(",") for l in r.stdout.splitlines() if l.startswith("PERFORMANCE,")][1]
    assert row[2] == "flat" and int(row[-2]) == 0  # failed_count against the rocSPARSE result


@pytest.mark.gpu
@pytest.mark.parametrize("pipeline", [1, 3])
def test_cli_multi_gpu_driver_on_one_gpu(cli, tmp_path, pipeline):
    """spmv-cli --gpus N (north_star: one process, one host thread per GPU, pinned staging per shard, ncclCommInitAll, local SpMV +
    exchange per step) with the one GPU this box has: the whole path runs -- communicator, per-thread stream, shard handle, the
    in-place allgather (pipeline 1) or the chunked step (pipeline 3) -- and the reference CLI's verdict on the gathered y passes.
    A request for more GPUs than the box has is refused."""
    rowptr, cols, vals = synth.random_csr(50000, 50000, 8, seed=13, kind="powerlaw")
    p = str(tmp_path / "m.bin2")
    write_bin2(p, 50000, 50000, rowptr, cols, vals)
    for strat in ("adaptive", "flat"):
        r = subprocess.run([cli, p, "-f", "bin2", "--gpus", "1", "--pipeline", str(pipeline), "--strategy", strat],
                           capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, (strat, r.stdout, r.stderr)
        assert "Congratulation, pass 50000 validation!" in r.stdout and f"gpus:1 pipeline:{pipeline}" in r.stdout
    import torch

    too_many = torch.cuda.device_count() + 1
    r = subprocess.run([cli, p, "-f", "bin2", "--gpus", str(too_many)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 2 and "visible" in r.stderr


@pytest.fixture(scope="module")
def mock_rccl(tmp_path_factory):
    """tests/cxx/mock_rccl.cpp built into a shared object: N ranks = N host threads of one process on ONE device (test infrastructure: RCCL
    itself refuses two ranks on one device, and no round had a box with more than one GPU)."""
    if not os.path.exists("/opt/rocm/bin/hipcc"):
        pytest.skip("hipcc not available")
    out = str(tmp_path_factory.mktemp("mock") / "libmock_rccl.so")
    subprocess.run(["/opt/rocm/bin/hipcc", "-O2", "-std=c++17", "-fPIC", "-shared", os.path.join(ROOT, "tests", "cxx", "mock_rccl.cpp"), "-o", out],
                   check=True, capture_output=True, timeout=600)
    return out


@pytest.mark.gpu
@pytest.mark.parametrize("ranks,pipeline", [(2, 1), (2, 3), (3, 1), (4, 4), (8, 2)])
def test_cli_several_ranks_on_one_gpu_over_a_mock_rccl(cli, mock_rccl, tmp_path, ranks, pipeline):
    """The N > 1 logic of the C / C++ sharded step, which one-rank RCCL runs cannot reach: spmv-cli --gpus N with N host threads on ONE device
    (SPMV_CLI_ONE_DEVICE) over the mock communicator (SPMV_ACC_RCCL_LIB): nnz-balanced shards of unequal row counts padded to one length, the
    in-place allgather at every rank's offset (pipeline 1), the chunked point-to-point fan-out with every rank paired with every other at every
    position (pipeline C: chunk sizes that do not divide the shards), spmv_acc_shard_prepare ahead of the first collective, and the reference
    CLI's verdict on rank 0's gathered y.  What the mock cannot show is RCCL itself (its transports, its stream semantics): see its header."""
    m = 60_000 + 7 * ranks
    rowptr, cols, vals = synth.random_csr(m, m, 9, seed=31 + ranks, kind="powerlaw")
    p = str(tmp_path / "m.bin2")
    write_bin2(p, m, m, rowptr, cols, vals)
    env = dict(os.environ, SPMV_ACC_RCCL_LIB=mock_rccl, SPMV_CLI_ONE_DEVICE="1", MOCK_RCCL_TIMEOUT_S="30")
    for strat in ("adaptive", "flat"):
        r = subprocess.run([cli, p, "-f", "bin2", "--gpus", str(ranks), "--pipeline", str(pipeline), "--strategy", strat], capture_output=True, text=True,
                           env=env, timeout=300)
        assert r.returncode == 0, (strat, r.stdout[-600:], r.stderr[-1200:])
        assert f"Congratulation, pass {m} validation!" in r.stdout and f"gpus:{ranks} pipeline:{pipeline}" in r.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("phase", ["prepare", "step"])
@pytest.mark.parametrize("pipeline", [1, 3])
def test_a_failing_rank_does_not_hang_its_peers(cli, mock_rccl, tmp_path, phase, pipeline):
    """a rank whose local SpMV fails must not leave its peers blocked in a collective.  Three ranks over the mock
    communicator; rank 1's call is sabotaged (a NULL x) in spmv_acc_shard_prepare or in its third step.  The failing rank still takes part in the
    step's exchanges (spmv_acc_shard_step), the ranks agree at the barrier behind the phase and all stop: the process ends by itself -- well inside
    the mock's own timeout, which would turn a missing peer into an error after 30 s -- with the driver's failure code and the failing rank named."""
    import time

    m = 40_000
    rowptr, cols, vals = synth.random_csr(m, m, 7, seed=77, kind="powerlaw")
    p = str(tmp_path / "m.bin2")
    write_bin2(p, m, m, rowptr, cols, vals)
    env = dict(os.environ, SPMV_ACC_RCCL_LIB=mock_rccl, SPMV_CLI_ONE_DEVICE="1", MOCK_RCCL_TIMEOUT_S="30", SPMV_CLI_FAIL=f"1:{phase}")
    t0 = time.time()
    r = subprocess.run([cli, p, "-f", "bin2", "--gpus", "3", "--pipeline", str(pipeline)], capture_output=True, text=True, env=env, timeout=200)
    took = time.time() - t0
    assert r.returncode == 5, (r.returncode, r.stdout[-400:], r.stderr[-800:])
    assert "rank 1: spmv_acc_shard_" + ("prepare" if phase == "prepare" else "step") in r.stderr
    assert "rank 0:" not in r.stderr and "rank 2:" not in r.stderr  # (the peers' exchanges all completed: nobody timed out waiting for rank 1)
    assert took < 25, took


# ---- readers pinned against the REFERENCE's own readers ----------------------------------------------------------------------------
def _read_dump(path):
    with open(path, "rb") as f:
        rows, cols, nnz, xlen = np.fromfile(f, dtype=np.int32, count=4)
        rp = np.fromfile(f, dtype=np.int32, count=rows + 1)
        ci = np.fromfile(f, dtype=np.int32, count=nnz)
        v = np.fromfile(f, dtype=np.float64, count=nnz)
        x = np.fromfile(f, dtype=np.float64, count=xlen)
    return int(rows), int(cols), int(nnz), rp, ci, v, x


def test_readers_match_reference_goldens(cli, tmp_path):
    """tests/golden/reader_cases.npz holds small input files and what the REFERENCE's readers (cli/csr_mtx_reader.hpp,
    csr_binary_reader.hpp, matrix_market_reader.hpp + sparse_format.h::to_csr) parsed from them; our readers must
    produce the same CSR bit for bit."""
    g = np.load(os.path.join(ROOT, "tests", "golden", "reader_cases.npz"))
    for name, fmt in zip(g["names"], g["formats"]):
        p = str(tmp_path / str(name))
        g[f"{name}__file"].tofile(p)
        out = str(tmp_path / (str(name) + ".dump"))
        r = subprocess.run([cli, p, "-f", str(fmt), "--dump-bin", out], capture_output=True, text=True)
        assert r.returncode == 0, (name, r.stderr)
        rows, cols, nnz, rp, ci, v, x = _read_dump(out)
        want_rows, want_cols, want_nnz = (int(t) for t in g[f"{name}__dims"])
        assert (rows, cols, nnz) == (want_rows, want_cols, want_nnz), name
        assert np.array_equal(rp, g[f"{name}__rowptr"]) and np.array_equal(ci, g[f"{name}__colidx"]), name
        assert np.array_equal(v, g[f"{name}__values"]), name  # same decimal -> same double
        if fmt == "csr":
            assert np.array_equal(x, g[f"{name}__x"]), name


def test_convert_bin2_writes_what_the_reference_converter_writes(cli, tmp_path, oracle):
    """`spmv-cli in.mtx -f mtx --convert-bin2 out.bin2` = the reference's Go converter for one file (tools/suitesparse-dl/conv/conv.go:92-150 +
    mm_parser.go:205-245, restated here in numpy): off-diagonals of symmetric / hermitian files mirrored, entries sorted by (row, column), little-endian
    header magic / version 2 / value type / rows / cols / nnz, rowptr, colindex, then no values (pattern), int32 (integer) or f64 (real; complex keeps
    the real part).  Byte for byte; and the reference's own compiled reader gets the same matrix back from the converted file (real / pattern: its
    integer branch overruns its buffer)."""
    import struct

    rng = np.random.default_rng(31)
    cases = [("real", "general"), ("real", "symmetric"), ("integer", "general"), ("pattern", "symmetric"), ("complex", "hermitian"), ("integer", "symmetric")]
    for k, (field, sym) in enumerate(cases):
        m = int(rng.integers(5, 120))
        n = m if sym != "general" else int(rng.integers(5, 150))
        pairs = set()
        while len(pairs) < 4 * m:
            r, c = int(rng.integers(1, m + 1)), int(rng.integers(1, n + 1))
            if sym != "general" and c > r:
                r, c = c, r  # lower triangle, as such files are stored
            pairs.add((r, c))
        pairs = sorted(pairs, key=lambda rc: (rc[1], rc[0]))  # column-major, as SuiteSparse files are
        vals = rng.integers(-50, 50, len(pairs)).astype(np.float64) if field == "integer" else rng.standard_normal(len(pairs))
        pm = str(tmp_path / f"c{k}.mtx")
        with open(pm, "w") as f:
            f.write(f"%%MatrixMarket matrix coordinate {field} {sym}\n% a comment\n{m} {n} {len(pairs)}\n")
            for (r, c), v in zip(pairs, vals):
                f.write({"pattern": f"{r} {c}\n", "integer": f"{r} {c} {int(v)}\n", "complex": f"{r} {c} {v:.17g} 0.25\n"}.get(field, f"{r} {c} {v:.17g}\n"))
        # conv.go, restated
        rows = [r - 1 for r, c in pairs] + [c - 1 for r, c in pairs if sym != "general" and r != c]
        cols = [c - 1 for r, c in pairs] + [r - 1 for r, c in pairs if sym != "general" and r != c]
        v = list(vals) + [x for (r, c), x in zip(pairs, vals) if sym != "general" and r != c]
        if field == "pattern":
            v = [1.0] * len(rows)
        order = sorted(range(len(rows)), key=lambda i: (rows[i], cols[i]))
        rows, cols, v = np.array(rows)[order], np.array(cols, dtype="<i4")[order], np.array(v)[order]
        rowptr = np.zeros(m + 1, dtype="<i4")
        np.add.at(rowptr, rows + 1, 1)
        rowptr = np.cumsum(rowptr).astype("<i4")
        valtype = {"pattern": 1, "integer": 2, "real": 3, "complex": 4}[field]
        body = b"" if valtype == 1 else (v.astype("<i4").tobytes() if valtype == 2 else v.astype("<f8").tobytes())
        want = struct.pack("<6i", 0x20211015, 2, valtype, m, n, len(rows)) + rowptr.tobytes() + cols.tobytes() + body
        pb = str(tmp_path / f"c{k}.bin2")
        r = subprocess.run([cli, pm, "-f", "mtx", "--convert-bin2", pb], capture_output=True, text=True)
        assert r.returncode == 0 and f"nnz={len(rows)} valtype={valtype}" in r.stdout, (field, sym, r.stdout, r.stderr)
        assert open(pb, "rb").read() == want, (field, sym)
        # read back: the same matrix as the .mtx gives (this CLI) ...
        d1, d2 = pm + ".dump", pb + ".dump"
        assert subprocess.run([cli, pm, "-f", "mtx", "--dump-bin", d1], capture_output=True).returncode == 0
        assert subprocess.run([cli, pb, "-f", "bin2", "--dump-bin", d2], capture_output=True).returncode == 0
        a, b = _read_dump(d1), _read_dump(d2)
        assert a[:3] == b[:3] and all(np.array_equal(p_, q_) for p_, q_ in zip(a[3:6], b[3:6])), (field, sym)
        # ... and as the reference's compiled reader gets from the converted file
        if oracle.ref_readers() is not None and valtype != 2:
            ref = oracle.ref_read_matrix(pb, "bin2")
            assert ref[:3] == b[:3] and all(np.array_equal(p_, q_) for p_, q_ in zip(ref[3:6], b[3:6])), (field, sym, "reference reader")
    # a .csr text file and a bin2 file convert too (real values); an unwritable target is an error, not a crash
    rowptr, cols, vals = synth.random_csr(40, 30, 4, seed=9, kind="uniform")
    pc = str(tmp_path / "t.csr")
    write_csr_text(pc, rowptr, cols, vals, rng.standard_normal(30))
    out = str(tmp_path / "t.bin2")
    assert subprocess.run([cli, pc, "-f", "csr", "--convert-bin2", out], capture_output=True).returncode == 0
    want = struct.pack("<6i", 0x20211015, 2, 3, 40, 30, int(rowptr[-1])) + rowptr.astype("<i4").tobytes() + cols.astype("<i4").tobytes() + vals.astype("<f8").tobytes()
    assert open(out, "rb").read() == want
    assert subprocess.run([cli, pc, "-f", "csr", "--convert-bin2", "/nonexistent-dir/x.bin2"], capture_output=True).returncode == 3


def test_readers_match_compiled_reference_random(cli, tmp_path, oracle):
    if oracle.ref_readers() is None:
        pytest.skip("oracle/_ref/libref_readers.so not built (no /root/reference on this machine)")
    rng = np.random.default_rng(123)
    for trial in range(6):
        m, n = int(rng.integers(1, 400)), int(rng.integers(1, 400))
        rowptr, cols, vals = synth.random_csr(m, n, int(rng.integers(1, 9)), seed=500 + trial,
                                              kind=["uniform", "powerlaw", "empty_rows"][trial % 3])
        x = rng.standard_normal(n)
        pc = str(tmp_path / f"t{trial}.csr")
        write_csr_text(pc, rowptr, cols, vals, x)
        pb = str(tmp_path / f"t{trial}.bin2")
        write_bin2(pb, m, n, rowptr, cols, vals)
        pm = str(tmp_path / f"t{trial}.mtx")
        A = {}
        for r in range(m):
            for j in range(rowptr[r], rowptr[r + 1]):
                A[(r + 1, int(cols[j]) + 1)] = float(vals[j])  # duplicates collapse: MatrixMarket entries are unique
        with open(pm, "w") as f:
            f.write("%%MatrixMarket matrix coordinate real general\n")
            f.write("%d %d %d\n" % (m, n, len(A)))
            for (r, c), val in A.items():
                f.write("%d %d %.17g\n" % (r, c, val))
        for path, fmt in ((pc, "csr"), (pb, "bin2"), (pm, "mtx")):
            want = oracle.ref_read_matrix(path, fmt)
            out = path + ".dump"
            rr = subprocess.run([cli, path, "-f", fmt, "--dump-bin", out], capture_output=True, text=True)
            assert rr.returncode == 0, rr.stderr
            got = _read_dump(out)
            assert got[:3] == want[:3], (trial, fmt)
            for a, b in zip(got[3:6], want[3:6]):
                assert np.array_equal(a, b), (trial, fmt)
            if fmt == "csr":
                assert np.array_equal(got[6], want[6])


def test_readers_reject_malformed_files_under_sanitizers(tmp_path):
    """The readers, compiled alone with AddressSanitizer + UBSan (CPU build), on corrupt inputs: every one must be
    refused with a message -- no sanitizer report, no crash, no out-of-range structure reaching the kernels.  (The
    reference's readers pass such files on: cli/csr_binary_reader.hpp:44-99 never checks the body against the header.)"""
    exe = tmp_path / "reader_sanitize"
    src = os.path.join(ROOT, "tests", "cxx", "reader_sanitize.cpp")
    subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                    "-I", os.path.join(ROOT, "spmv_acc_amd", "cli"), src, "-o", str(exe)], check=True)
    rowptr = np.array([0, 2, 3, 5], dtype=np.int32)
    cols = np.array([0, 2, 1, 0, 2], dtype=np.int32)
    vals = np.arange(1.0, 6.0)
    hdr = lambda rows, ncols, nnz, vt=3, magic=0x20211015, ver=2: struct.pack("<6i", magic, ver, vt, rows, ncols, nnz)
    body = rowptr.tobytes() + cols.tobytes() + vals.tobytes()
    files = {
        "good.bin2": (hdr(3, 3, 5) + body, "ok 3 3 5 5"),
        "bad_magic.bin2": (hdr(3, 3, 5, magic=0x1234) + body, "error"),
        "bad_version.bin2": (hdr(3, 3, 5, ver=1) + body, "error"),
        "bad_valtype.bin2": (hdr(3, 3, 5, vt=9) + body, "error"),
        "short_header.bin2": (hdr(3, 3, 5)[:10], "error"),
        "truncated.bin2": (hdr(3, 3, 5) + body[:-9], "error"),
        "huge_nnz.bin2": (hdr(3, 3, 2**31 - 1) + body, "error"),       # must not allocate 24 GB first
        "huge_rows.bin2": (hdr(2**31 - 1, 3, 5) + body, "error"),
        "negative.bin2": (hdr(-3, 3, 5) + body, "error"),
        "col_out_of_range.bin2": (hdr(3, 2, 5) + body, "error"),          # column 2 with 2 columns
        "negative_col.bin2": (hdr(3, 3, 5) + rowptr.tobytes() + np.array([0, -1, 1, 0, 2], np.int32).tobytes() + vals.tobytes(), "error"),
        "rowptr_decreases.bin2": (hdr(3, 3, 5) + np.array([0, 3, 2, 5], np.int32).tobytes() + cols.tobytes() + vals.tobytes(), "error"),
        "rowptr_overshoots.bin2": (hdr(3, 3, 5) + np.array([0, 2, 9, 5], np.int32).tobytes() + cols.tobytes() + vals.tobytes(), "error"),
        "rowptr_nnz_mismatch.bin2": (hdr(3, 3, 5) + np.array([0, 2, 3, 4], np.int32).tobytes() + cols.tobytes() + vals.tobytes(), "error"),
        "pattern.bin2": (hdr(3, 3, 5, vt=1) + rowptr.tobytes() + cols.tobytes(), "ok 3 3 5 5"),
        "good.mtx": (b"%%MatrixMarket matrix coordinate real general\n% c\n3 3 2\n1 1 1.5\n3 2 -2\n", "ok 3 3 2 1"),
        "index_zero.mtx": (b"%%MatrixMarket matrix coordinate real general\n3 3 1\n0 1 1.5\n", "error"),
        "index_big.mtx": (b"%%MatrixMarket matrix coordinate real general\n3 3 1\n1 4 1.5\n", "error"),
        "missing_value.mtx": (b"%%MatrixMarket matrix coordinate real general\n3 3 1\n1 1\n", "error"),
        "too_few.mtx": (b"%%MatrixMarket matrix coordinate real general\n3 3 2\n1 1 1.0\n", "error"),
        "too_many.mtx": (b"%%MatrixMarket matrix coordinate real general\n3 3 1\n1 1 1.0\n2 2 1.0\n", "error"),
        "huge_declared.mtx": (b"%%MatrixMarket matrix coordinate real general\n3 3 999999999999999999\n1 1 1.0\n", "error"),
        "huge_dims.mtx": (b"%%MatrixMarket matrix coordinate real general\n99999999999 3 1\n1 1 1.0\n", "error"),
        "negative_dims.mtx": (b"%%MatrixMarket matrix coordinate real general\n-3 3 1\n1 1 1.0\n", "error"),
        "array_format.mtx": (b"%%MatrixMarket matrix array real general\n2 2\n1\n2\n3\n4\n", "error"),
        "garbage.mtx": (bytes(range(256)) * 8, "error"),
        "empty.mtx": (b"", "error"),
        "good.csr": (b"hdr\n1 2 3\n0 1 0\n0 2 3\n0.5 0.25\n", "ok 2 2 3 1"),
        "four_lines.csr": (b"hdr\n1 2 3\n0 1 0\n0 2 3\n", "error"),
        "col_out_of_range.csr": (b"hdr\n1 2 3\n0 5 0\n0 2 3\n0.5 0.25\n", "error"),
        "rowptr_mismatch.csr": (b"hdr\n1 2 3\n0 1 0\n0 2 4\n0.5 0.25\n", "error"),
        "rowptr_decreases.csr": (b"hdr\n1 2 3\n0 1 0\n0 3 2 3\n0.5 0.25\n", "error"),
        "bad_token.csr": (b"hdr\n1 x 3\n0 1 0\n0 2 3\n0.5 0.25\n", "error"),
        "empty.csr": (b"", "error"),
    }
    paths = []
    for name, (data, _) in files.items():
        p = tmp_path / name
        p.write_bytes(data)
        paths.append(str(p))
    r = subprocess.run([str(exe)] + paths, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "Sanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-2000:]
    lines = r.stdout.strip().splitlines()
    assert len(lines) == len(files), r.stdout
    for (name, (_, want)), got in zip(files.items(), lines):
        assert got.startswith(want), (name, got)
