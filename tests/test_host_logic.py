"""CPU suite: the product library loads without a GPU, exports every C-ABI symbol include/spmv_acc.h
declares, and its host-side logic (strategy names, pickers, host form of the row-block preprocessing
pass, row partition) matches the oracle / the reference-generated goldens.  No compute calls here."""
import os
import re
import subprocess
import sys

import numpy as np
import pytest

import spmv_acc_amd
from spmv_acc_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


def test_library_exports_every_declared_symbol(hiplib):
    header = open(os.path.join(ROOT, "include", "spmv_acc.h")).read()
    # names followed by '(' at declaration level
    declared = set(re.findall(r"\b(sparse_spmv|spmv_acc_[a-z_0-9]+)\s*\(", header))
    assert declared == set(spmv_acc_amd.C_ABI_SYMBOLS), declared ^ set(spmv_acc_amd.C_ABI_SYMBOLS)
    for s in declared:
        assert hasattr(hiplib, s), s


def test_cxx_symbols_for_cli_and_benchmark(hiplib):
    """The mangled C++ entry points spmv-cli / spmv-gpu-benchmark link against."""
    out = subprocess.run(["nm", "-D", "--defined-only", "-C", spmv_acc_amd.LIB_PATH], capture_output=True,
                         text=True, check=True).stdout
    for name in ("sparse_csr_spmv(int, double, double, csr_desc<int, double>, csr_desc<int, double>",
                 "sparse_spmv(int, double, double, int, int, int const*", "default_sparse_spmv(",
                 "adaptive_sparse_spmv(", "flat_sparse_spmv(", "line_enhance_sparse_spmv(",
                 "adaptive_enhance_sparse_spmv(", "adaptive_line_sparse_spmv(", "vec_row_sparse_spmv(",
                 "adaptive_vec_row_sparse_spmv(", "adaptive_flat_sparse_spmv(", "thread_row_sparse_spmv(",
                 "wf_row_sparse_spmv(", "light_sparse_spmv(", "block_row_sparse_spmv(",
                 "void csr_adaptive_plus_sparse_spmv<true, int, double>(SpMVAccHanele*",
                 "void csr_adaptive_plus_sparse_spmv<false, int, double>(SpMVAccHanele*"):
        assert name in out, name


def test_strategy_names_follow_reference_matching(hiplib):
    P = lambda s: hiplib.spmv_acc_parse_strategy(s.encode())
    names = [hiplib.spmv_acc_strategy_name(i).decode() for i in range(11)]
    assert tuple(names) == spmv_acc_amd.STRATEGIES
    for i, n in enumerate(names):
        assert P(n) == i and P(n.upper()) == i
    assert P("LINE_ENHANCE") == 7 and P("LINE") == 8  # line_enhance is tested before line
    assert P("my_default_build") == 0  # regex MATCHES = substring (configure.cmake:18-37)
    assert P("nonsense") == -1
    assert hiplib.spmv_acc_set_strategy(b"nonsense") == -1
    hiplib.spmv_acc_clear_error()


def test_strategy_env_and_setter():
    code = ("import spmv_acc_amd as s; print(s.get_strategy()); s.set_strategy('FLAT'); print(s.get_strategy())")
    env = dict(os.environ, SPMV_ACC_KERNEL_STRATEGY="line_enhance", PYTHONPATH=ROOT)
    out = subprocess.run(["python", "-c", code], env=env, capture_output=True, text=True, check=True).stdout.split()
    assert out == ["line_enhance", "flat"]
    env.pop("SPMV_ACC_KERNEL_STRATEGY")
    out = subprocess.run(["python", "-c", code], env=env, capture_output=True, text=True, check=True).stdout.split()
    assert out == ["adaptive", "flat"]  # build-time default of this tree: KERNEL_STRATEGY_ADAPTIVE


def test_adaptive_branch_matches_oracle(hiplib, oracle):
    rng = np.random.default_rng(3)
    seen = set()
    for trial in range(300):
        m = int(rng.integers(1, 5000))
        kind = trial % 5
        lens = rng.integers(0, [3, 9, 60, 9, 9][kind], m).astype(np.int64)
        if kind == 3:
            lens[: m // 2] *= 7  # heavy first half
        if kind == 4:
            lens[: m // 2] = 0  # empty first half: the reference would divide by zero here
        scale = [1, 1, 1, 1, 1][kind]
        rp = np.zeros(m + 1, dtype=np.int64)
        np.cumsum(lens * scale, out=rp[1:])
        # blow nnz up so every threshold (0xC00000, 2^23) is crossed by some trials
        mult = int(rng.choice([1, 1, 2000, 20000]))
        rp = np.minimum(rp * mult, 2**31 - 2).astype(np.int32)
        got = spmv_acc_amd.adaptive_branch(m, rp)
        assert got == oracle.adaptive_pick(rp), (m, kind, mult)
        seen.add(got)
    assert seen >= {1, 2, 3, 4}


def test_plus_analysis_matches_reference_goldens(hiplib):
    g = np.load(os.path.join(GOLD, "analysis_cases.npz"))
    for name in g["names"]:
        rp = g[f"{name}__rowptr"]
        m = rp.size - 1
        for k, (threads, vec, min_nnz) in enumerate(g["params"]):
            blocks, bp, fbr = spmv_acc_amd.adaptive_plus_analyze(rp, m, int(min_nnz), int(threads), int(vec))
            assert np.array_equal(bp, g[f"{name}__{k}__bp"]), (name, k)
            assert np.array_equal(fbr, g[f"{name}__{k}__fbr"]), (name, k)


def test_plus_analysis_matches_oracle_random(hiplib, oracle):
    rng = np.random.default_rng(5)
    for trial in range(150):
        m = int(rng.integers(1, 3000))
        kind = trial % 4
        if kind == 0:
            lens = rng.integers(0, 12, m)
        elif kind == 1:
            lens = np.minimum((rng.pareto(1.2, m) * 3).astype(np.int64), 20000)
        elif kind == 2:
            lens = rng.integers(0, 3, m)
            lens[rng.integers(0, m, 3)] = rng.integers(2000, 30000, 3)
        else:
            lens = rng.integers(0, 700, m)
        rp = np.zeros(m + 1, dtype=np.int32)
        np.cumsum(lens, out=rp[1:])
        for threads, vec, min_nnz in ((512, 1, 2048), (512, 8, 2048), (256, 2, 1024), (1024, 32, 4096)):
            a = spmv_acc_amd.adaptive_plus_analyze(rp, m, min_nnz, threads, vec)
            b = oracle.adaptive_plus_analyze(rp, min_nnz, threads, vec)
            assert a[0] == b[0] and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2]), (trial, vec)
            if oracle.ref() is not None:
                c = oracle.ref_adaptive_plus_analyze(rp, min_nnz, threads, vec)
                assert a[0] == c[0] and np.array_equal(a[1], c[1]) and np.array_equal(a[2], c[2])


def test_plus_vec_and_bp_len(hiplib, oracle):
    L = oracle.lib()
    for m, nnz in ((10, 0), (10, 20), (10, 21), (100, 450), (7, 7 * 64), (7, 7 * 65), (8_217_820, 40_451_632)):
        assert hiplib.spmv_acc_adaptive_plus_vec(m, nnz) == L.oracle_adaptive_plus_vec(m, nnz)
    for nnz in (0, 1, 1023, 1024, 1025, 40_451_632):
        for s in (1024, 2048):
            assert hiplib.spmv_acc_break_points_len(nnz, s) == L.oracle_break_points_len(nnz, s)


def test_partition_rows(hiplib):
    rowptr, _, _ = synth.random_csr(10_000, 10_000, 8, seed=9, kind="powerlaw")
    eq = spmv_acc_amd.partition_rows(10_000, 8, mode=0)
    assert eq[0] == 0 and eq[-1] == 10_000 and np.all(np.diff(eq) == 1250)
    eq = spmv_acc_amd.partition_rows(10_001, 8, mode=0)
    assert eq[-1] == 10_001 and np.all(np.diff(eq)[:-1] == 1251) and np.diff(eq)[-1] <= 1251
    bal = spmv_acc_amd.partition_rows(10_000, 8, mode=1, h_rowptr=rowptr)
    assert bal[0] == 0 and bal[-1] == 10_000 and np.all(np.diff(bal) >= 0)
    shares = np.diff(rowptr[bal].astype(np.int64))
    longest = int(np.diff(rowptr).max())
    assert shares.max() - shares.min() <= 2 * longest + 1  # balanced up to one row at each cut


def test_no_cpu_fallback_for_compute(hiplib):
    """Host arrays are refused: the product path never computes on the CPU."""
    import torch

    rowptr, cols, vals = synth.random_csr(16, 16, 3, seed=1)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a))
    with pytest.raises(spmv_acc_amd.SpmvAccError):
        spmv_acc_amd.csr_spmv(1.0, 1.0, 16, 16, int(rowptr[-1]), t(rowptr), t(cols), t(vals), torch.zeros(16, dtype=torch.float64),
                              torch.zeros(16, dtype=torch.float64))


def test_missing_library_fails_loudly(tmp_path):
    with pytest.raises(spmv_acc_amd.SpmvAccError, match="no CPU fallback"):
        spmv_acc_amd.load_library(str(tmp_path / "libspmv_acc.so"))


def test_product_never_touches_oracle():
    """Nothing under spmv_acc_amd/ or include/ may import, link or execute anything under oracle/."""
    bad = []
    for base in ("spmv_acc_amd", "include"):
        for dirpath, _, files in os.walk(os.path.join(ROOT, base)):
            if "build" in dirpath.split(os.sep):
                continue
            for f in files:
                if f.endswith((".py", ".cpp", ".hpp", ".h", ".hip", "Makefile")):
                    text = open(os.path.join(dirpath, f), errors="ignore").read()
                    if re.search(r"oracle_lib|liboracle|libref_analyze|#include\s+\"[^\"]*oracle|import oracle|from oracle", text):
                        bad.append(os.path.join(dirpath, f))
    assert not bad, bad


def test_env_tunables_seed_defaults():
    """SPMV_ACC_TUNABLES seeds tunables for processes that cannot call spmv_acc_set_tunable (the reference's own
    executables linked against the library); reset returns to the seeded values; unknown names are ignored."""
    code = ("import spmv_acc_amd as s; l = s.load_library(); "
            "print(l.spmv_acc_get_tunable(b'validate'), l.spmv_acc_get_tunable(b'flat_finish'), l.spmv_acc_get_tunable(b'xcd_chunk')); "
            "l.spmv_acc_set_tunable(b'validate', 0); l.spmv_acc_reset_tunables(); print(l.spmv_acc_get_tunable(b'validate'))")
    env = dict(os.environ, SPMV_ACC_TUNABLES="validate=1,flat_finish=0,no_such=5", PYTHONPATH=ROOT)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stderr[-1000:]
    assert r.stdout.split() == ["1", "0", "16", "1"], r.stdout


def test_c_header_is_plain_c(tmp_path):
    """include/spmv_acc.h is the FFI surface: it must compile as C99 (and C++11) with nothing but the standard headers, and
    every prototype it declares must link against the library (the loader test checks the symbols, this one the declarations)."""
    src = tmp_path / "use_header.c"
    src.write_text('#include "spmv_acc.h"\n'
                   'int main(void) { int out[9]; (void)out; return spmv_acc_break_points_len(4096, 1024) == 5 ? 0 : 1; }\n')
    inc = os.path.join(ROOT, "include")
    for cmd in (["gcc", "-std=c99"], ["g++", "-std=c++11", "-x", "c++"]):
        subprocess.run(cmd + ["-Wall", "-Wextra", "-pedantic", "-Werror", "-fsyntax-only", "-I", inc, str(src)], check=True)


def test_library_links_only_the_hip_runtime(hiplib):
    """north_star: 'not a wrapper over rocSPARSE'.  The shared library's NEEDED entries are the HIP runtime and the C/C++
    runtimes, nothing else (rocPRIM, used by the device-side row-block analysis, is header-only)."""
    lib = os.path.join(ROOT, "spmv_acc_amd", "lib", "libspmv_acc.so")
    out = subprocess.run(["readelf", "-d", lib], capture_output=True, text=True, check=True).stdout
    needed = re.findall(r"NEEDED\)\s+Shared library: \[([^\]]+)\]", out)
    allowed = ("libamdhip64", "libstdc++", "libm.", "libgcc_s", "libc.", "ld-linux", "libdl", "libpthread", "librt")
    assert needed and all(n.startswith(allowed) for n in needed), needed


def test_cmake_build_exports_the_same_symbols(tmp_path):
    """The top-level CMakeLists.txt (for consumers that add_subdirectory() this repository, as the reference's CMake would) builds
    target `spmv-acc-kernels` = libspmv_acc.so with the same exported C ABI as the Makefile build."""
    import shutil

    if not shutil.which("cmake") or not os.path.exists("/opt/rocm/bin/hipcc"):
        pytest.skip("cmake / hipcc not available")
    gen = ["-G", "Ninja"] if shutil.which("ninja") else []
    subprocess.run(["cmake", "-S", ROOT, "-B", str(tmp_path)] + gen + ["-DCMAKE_CXX_COMPILER=/opt/rocm/bin/hipcc",
                   "-DCMAKE_BUILD_TYPE=Release", "-DKERNEL_STRATEGY=flat"], check=True, capture_output=True)
    subprocess.run(["cmake", "--build", str(tmp_path), "-j", "8"], check=True, capture_output=True)
    out = subprocess.run(["nm", "-D", "--defined-only", str(tmp_path / "libspmv_acc.so")], capture_output=True, text=True, check=True).stdout
    exported = {line.split()[-1] for line in out.splitlines() if line.strip()}
    missing = [s for s in spmv_acc_amd.C_ABI_SYMBOLS if s not in exported]
    assert not missing, missing


def test_cmake_accepts_the_reference_cache_variables(tmp_path):
    """A configure line written for the reference (config.cmake:2-51) configures this tree: every option is accepted,
    KERNEL_STRATEGY is matched by case-insensitive substring in the reference's order (src/configure.cmake:17-40: any string
    containing a name selects it, `line_enhance` before `line`), and an unknown strategy is the same fatal error."""
    import shutil

    if not shutil.which("cmake") or not os.path.exists("/opt/rocm/bin/hipcc"):
        pytest.skip("cmake / hipcc not available")
    common = ["-DCMAKE_CXX_COMPILER=/opt/rocm/bin/hipcc", "-DHIP_ENABLE_FLAG=ON", "-DSPMV_BUILD_TOOLS=OFF", "-DSPMV_BUILD_BENCHMARK=ON",
              "-DSPMV_OMP_ENABLED_FLAG=ON", "-DDEVICE_SIDE_VERIFY_FLAG=ON", "-DBENCHMARK_CUDA_ENABLE_FLAG=OFF",
              "-DBENCHMARK_FORCE_SYNC_KERNELS=ON", "-DAVAILABLE_CU=256", "-DWAVEFRONT_SIZE=64", "-DWF_REDUCE=LDS",
              "-DFLAT_SEGMENT_SUM_REDUCE=ON"]
    for given, macro in (("Flat", "KERNEL_STRATEGY_FLAT"), ("my_line_enhance_build", "KERNEL_STRATEGY_LINE_ENHANCE"),
                         ("LINE", "KERNEL_STRATEGY_LINE"), ("wf_row", "KERNEL_STRATEGY_WAVEFRONT_ROW")):
        b = tmp_path / given
        out = subprocess.run(["cmake", "-S", ROOT, "-B", str(b), f"-DKERNEL_STRATEGY={given}"] + common, capture_output=True, text=True)
        assert out.returncode == 0, out.stderr[-600:]
        assert f"({macro})" in out.stdout, out.stdout[-400:]
    out = subprocess.run(["cmake", "-S", ROOT, "-B", str(tmp_path / "bad"), "-DKERNEL_STRATEGY=csr5"] + common, capture_output=True, text=True)
    assert out.returncode != 0 and "unsupported kernel strategy" in out.stderr
    out = subprocess.run(["cmake", "-S", ROOT, "-B", str(tmp_path / "bad2"), "-DWF_REDUCE=tree"] + common[:-2], capture_output=True, text=True)
    assert out.returncode != 0 and "unsupported wavefront reduction strategy" in out.stderr


def test_round3_switches_and_entry_points_without_a_gpu(hiplib, tmp_path):
    """Host-side behaviour of the round-3 surface that needs no device: SPMV_ACC_DETERMINISTIC seeds the tunable; the tune-cache
    setter accepts a path / None; the library stream is per host thread; the shard handle API refuses bad arguments before it
    looks for RCCL or a device; the all-plans stale check is a no-op on an empty cache."""
    import ctypes
    import threading

    code = "import spmv_acc_amd as s; l = s.load_library(); print(l.spmv_acc_get_tunable(b'deterministic'), l.spmv_acc_get_tunable(b'col_slabs'))"
    for env_val, want in (("1", "1"), ("0", "0")):
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300,
                           env=dict(os.environ, SPMV_ACC_DETERMINISTIC=env_val, PYTHONPATH=ROOT))
        assert r.returncode == 0 and r.stdout.split() == [want, "-1"], (r.stdout, r.stderr[-500:])  # (col_slabs: automatic since round 6)
    # the late round-3 switches and their shipped values: the full row-pointer check is opt-in, the slab passes are automatic (-1),
    # the slab-major copy is opt-in; SPMV_ACC_TUNABLES seeds any of them for a process that cannot call the setter
    assert [hiplib.spmv_acc_get_tunable(n) for n in (b"guard_full", b"slab_segments", b"col_slabs", b"col16", b"rowblock_target")] == [0, -1, -1, -1, -1]
    assert hiplib.spmv_acc_query_plan_slab_passes(None, 5) == -2  # no such plan
    assert hiplib.spmv_acc_query_plan_settled(None, 5) == -2
    code = "import spmv_acc_amd as s; l = s.load_library(); print(l.spmv_acc_get_tunable(b'slab_segments'), l.spmv_acc_get_tunable(b'guard_full'))"
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300,
                       env=dict(os.environ, SPMV_ACC_TUNABLES="slab_segments=0,guard_full=1", PYTHONPATH=ROOT))
    assert r.returncode == 0 and r.stdout.split() == ["0", "1"], (r.stdout, r.stderr[-500:])
    spmv_acc_amd.set_tune_cache(str(tmp_path / "tune.txt"))
    spmv_acc_amd.set_tune_cache(None)
    assert hiplib.spmv_acc_check_plans() == 0 and hiplib.spmv_acc_cached_plans() == 0
    # stream: set here, invisible to another thread, restored
    hiplib.spmv_acc_set_stream(ctypes.c_void_p(0x1234))
    seen = []
    t = threading.Thread(target=lambda: seen.append(hiplib.spmv_acc_get_stream()))
    t.start()
    t.join()
    assert hiplib.spmv_acc_get_stream() == 0x1234 and seen == [None]
    hiplib.spmv_acc_set_stream(None)
    shard = ctypes.c_void_p()
    dummy = ctypes.c_void_p(0x10)
    assert hiplib.spmv_acc_shard_create(ctypes.byref(shard), None, 1, 10, 10, 10, 5, dummy, dummy, dummy, 1) == 2  # no communicator
    assert hiplib.spmv_acc_shard_create(ctypes.byref(shard), dummy, 1, 10, 9, 10, 5, dummy, dummy, dummy, 1) == 2  # pad < rows
    assert hiplib.spmv_acc_shard_create(None, dummy, 1, 10, 10, 10, 5, dummy, dummy, dummy, 1) == 2
    assert hiplib.spmv_acc_shard_step(None, 1.0, 0.0, dummy, None, dummy) == 2 and hiplib.spmv_acc_shard_destroy(None) == 0
    assert hiplib.spmv_acc_shard_pipeline(None) == 0
    hiplib.spmv_acc_clear_error()


def test_reference_flat_benchmark_tu_builds_against_include(hiplib, tmp_path):
    """The drop-in pinned with the reference's OWN translation unit: benchmark/flat/spmv_acc_flat.cpp (the benchmark's private flat copy: it
    launches pre_calc_break_point / pre_calc_break_point_v2 itself and expands FLAT_KERNEL_WRAPPER / FLAT_KERNEL_ONE_PASS_WRAPPER,
    spmv_acc_flat.cpp:14-15,34,38,65-71) and benchmark/utils/benchmark_time.cpp are compiled UNMODIFIED, from where they lie, with this
    repository's include/ standing where the reference's src/acc would (benchmark_spmv_acc.hpp:14-25 names the same header paths), and linked
    with libspmv_acc.so and a main that names flat_sparse_spmv<V1>, <V2>, adaptive_flat_sparse_spmv<V1> and segment_sum_flat_sparse_spmv.
    Runs where /root/reference exists (this container); nothing reference-derived travels to the GPU box."""
    ref = "/root/reference"
    tu = os.path.join(ref, "benchmark", "flat", "spmv_acc_flat.cpp")
    if not os.path.exists(tu) or not os.path.exists("/opt/rocm/bin/hipcc"):
        pytest.skip("the reference tree (or hipcc) is not on this machine")
    libdir = os.path.dirname(spmv_acc_amd.LIB_PATH)
    exe = str(tmp_path / "ref_flat_tu")
    cmd = ["/opt/rocm/bin/hipcc", "-w", "-std=c++17", "--offload-arch=gfx950", "-I", os.path.join(ROOT, "include"),
           "-I", os.path.join(ref, "benchmark"), "-I", os.path.join(ref, "cli"),
           os.path.join(ROOT, "tests", "cxx", "ref_flat_tu_main.cpp"), tu, os.path.join(ref, "benchmark", "utils", "benchmark_time.cpp"),
           "-L", libdir, "-lspmv_acc", f"-Wl,-rpath,{libdir}", "-o", exe]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    # every undefined symbol of the executable that belongs to this library is one the library defines (the link above already failed otherwise);
    # what the reference TU needs from us at link time is the strategy entry its launch macros forward to
    undef = subprocess.run(["nm", "-u", "-C", exe], capture_output=True, text=True, check=True).stdout
    assert "spmv_acc_csr_spmv_strategy" in undef
    defined = subprocess.run(["nm", "-C", "--defined-only", exe], capture_output=True, text=True, check=True).stdout
    for name in ("flat_sparse_spmv<1>", "flat_sparse_spmv<2>", "adaptive_flat_sparse_spmv<1>", "adaptive_flat_sparse_spmv<2>",
                 "segment_sum_flat_sparse_spmv", "BenchmarkTime::set_time"):
        assert name in defined, name
    # it loads and runs to main() without a GPU (the SpMV calls sit behind a false test)
    assert subprocess.run([exe], capture_output=True, timeout=60).returncode == 0


def test_every_size_threshold_names_a_test():
    """tests/size_thresholds.py: every size-selected branch of the engine is registered with the test(s) that cross it; every named constant of the
    engine's sources is either such a rule or listed as geometry.  (Round 4's wrong-result regression sat behind a size rule no test crossed.)"""
    import glob
    import re

    import size_thresholds as st

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    test_text = "\n".join(open(f).read() for f in glob.glob(os.path.join(root, "tests", "test_*.py")))
    defined = set(re.findall(r"^def (test_\w+)\(", test_text, flags=re.M))
    registered = set()
    for name, path, pattern, tests, what in st.SIZE_RULES:
        src = open(os.path.join(root, path)).read()
        assert re.search(pattern, src), f"{name}: the rule is no longer where the registry says ({path}: {pattern}) -- re-register it"
        assert tests, name
        for t in tests:
            assert t in defined, f"{name}: names the test {t}, which does not exist"
        registered.update(re.findall(r"k[A-Z]\w+", name))
    found = {}
    for path in st.SCANNED:
        for const in re.findall(r"constexpr\s+[\w:<> ]+?\s+(k[A-Z]\w*)\s*=", open(os.path.join(root, path)).read()):
            found[const] = path
    unknown = sorted(c for c in found if c not in registered and c not in st.NOT_SIZE_RULES)
    assert not unknown, f"constants neither registered as size rules nor listed as geometry in tests/size_thresholds.py: {[(c, found[c]) for c in unknown]}"
    stale = sorted(c for c in st.NOT_SIZE_RULES if c not in found)
    assert not stale, f"tests/size_thresholds.py lists constants that no longer exist: {stale}"


def test_row_block_kernel_instances_fit_eight_waves_per_simd(tmp_path):
    """Register budget and wait counts of the kernel every FEM-class stand-in and the headline settle on.  hipcc's own resource remarks and assembly
    for k_rowblock.hip, no GPU needed.
    (1) Registers (round 5's review found the headline instance at 65 VGPRs since a store flavour changed: 72 allocated, 7 waves per SIMD instead of 8,
    unnoticed): every instance -- plain colindex, row digest, 16-bit columns, and since the staging's tail body is a non-unrolled loop the hinted ones
    too (65-72 before) -- stays within 64 VGPRs (8 waves per SIMD, the most gfx950 runs), none spills.
    (2) Waits (round 6, profiles/r06_col16_counters.md section 5): with per-step wave-uniform branches around the staging's loads the compiler's
    waitcnt pass placed vmcnt(0) in front of the first 16-bit column decode -- every value load back before the first gather left -- and vmcnt(2) in
    front of the colindex gathers; the all-steps body has no such branch and must show vmcnt(5) (first step's record / offsets or colindex back, five
    later stream loads in flight) followed by vmcnt(8) (second step, with the first step's four gathers in flight behind it)."""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "tools"))
    import resource_table

    csrc = os.path.join(root, "spmv_acc_amd", "csrc")
    r = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-DKERNEL_STRATEGY_ADAPTIVE", "-I" + os.path.join(root, "include"),
                        "-Rpass-analysis=kernel-resource-usage", "--cuda-device-only", "-S", os.path.join(csrc, "k_rowblock.hip"), "-o", str(tmp_path / "k_rowblock.s")],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    rows = [k for k in resource_table.parse(r.stderr) if k["name"].startswith("rowblock_stream_kernel<")]
    assert len(rows) >= 100, len(rows)
    for k in rows:
        assert k["scratch"] == 0, k
        assert k["agprs"] == 0, k
        assert k["vgprs"] <= 64, k
        assert k["occupancy"] >= 8, k
    asm = open(tmp_path / "k_rowblock.s").read()
    checked = 0
    for k in rows:
        args = [a.strip() for a in k["name"][len("rowblock_stream_kernel<"):-1].split(",")]
        if args[4] == "true":
            continue  # (hinted gathers are buffer loads with their own address set-up between the waits)
        body = asm[asm.index("\n" + k["mangled"] + ":"):]
        body = body[:body.index("s_endpgm")]
        lines = body.split("\n")
        ok = False
        for i, ln in enumerate(lines):
            if "s_waitcnt vmcnt(5)" not in ln:
                continue
            gathers = 0
            for nxt in lines[i + 1:i + 300]:
                if re.search(r"global_load_dwordx2 [^\n]*, s\[", nxt):
                    gathers += 1
                elif "s_waitcnt vmcnt(8)" in nxt:
                    ok = ok or gathers == 4  # the first step's four gathers left before the second step's columns were waited for
                    break
                elif "s_waitcnt vmcnt(" in nxt and gathers:
                    break
        assert ok, (k["name"], re.findall(r"s_waitcnt vmcnt\((\d+)\)", body))
        checked += 1
    assert checked >= 60, checked


def test_tunable_table_stays_small_and_documented():
    """the engine's state space is what has to be tested, so the tunable table is capped (45 in round 5, 35 since round 6: ten A/B
    switches whose sweeps had flat-lined became constants) and every entry is documented in INTEGRATION.md's table under its name and default."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = open(os.path.join(root, "spmv_acc_amd", "csrc", "config.cpp")).read()
    table = src[src.index("Tunable g_tunables[] = {"):src.index("static_assert(sizeof(g_tunables)")]
    entries = re.findall(r'^\s*\{"(\w+)",\s*([^,]+),', table, flags=re.M)
    names = [n for n, _ in entries]
    assert len(names) == len(set(names)) and len(names) <= 35, (len(names), names)
    for gone in ("xcd_remap", "xcd_chunk_tiles", "copy_nt", "stage_fast", "early_y", "rescue_flat", "plus_ref_vec", "tune_protocol", "legacy_kernels", "vector_target"):
        assert gone not in names, gone
    enum = src_enum = open(os.path.join(root, "spmv_acc_amd", "csrc", "engine_internal.hpp")).read()
    ids = re.findall(r"kT_(\w+)", enum[enum.index("enum TunableId {"):enum.index("kTunableCount")])
    assert ids == names, "TunableId and the table must list the same names in the same order"
    doc = open(os.path.join(root, "INTEGRATION.md")).read()
    for n, default in entries:
        assert re.search(r"^\| `%s` \| " % re.escape(n), doc, flags=re.M), f"INTEGRATION.md's tunable table lacks `{n}`"
    listed = re.findall(r"^\| `(\w+)` \| [^|]+ \| ", doc[doc.index("| name | default | meaning |"):].split("\n\n")[0], flags=re.M)
    assert [n for n in listed if n not in names] == [], "INTEGRATION.md documents tunables that no longer exist"
