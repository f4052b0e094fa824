"""Synthetic CSR matrices for tests and bench.py (there is no network: the SuiteSparse files the
reference's scripts name -- examples/large-data-set-batch.sh:24-52 -- are not available, and its
examples/data/*.csr are Git-LFS stubs).  Stand-ins follow SURVEY.md 8(d): same (rows, cols, nnz) and a
structure of the same kind; every generator is seeded and deterministic.

Small matrices (tests, CPU) are built with numpy; the BASELINE.json-size ones are built with torch on
the GPU so nothing large crosses PCIe.
"""
from __future__ import annotations

import numpy as np

# (rows, cols, nnz) of the sweep the reference's script carries -- examples/large-data-set-batch.sh:24-52
LARGE_SET = {
    "boneS10": (914_898, 914_898, 28_191_660),
    "Bump_2911": (2_911_419, 2_911_419, 65_320_659),
    "Cube_Coup_dt6": (2_164_760, 2_164_760, 64_685_452),
    "dielFilterV3real": (1_102_824, 1_102_824, 45_204_422),
    "Ga41As41H72": (268_096, 268_096, 9_378_286),
    "Hardesty3": (8_217_820, 7_591_564, 40_451_632),
    "largebasis": (440_020, 440_020, 5_560_100),
    "RM07R": (381_689, 381_689, 37_464_962),
    "TSOPF_RS_b2383": (38_120, 38_120, 16_171_169),
    "vas_stokes_2M": (2_146_677, 2_146_677, 65_129_037),
}
# the two endpoints BASELINE.json configs[2] names; not in the reference's scripts (dims as listed by SuiteSparse)
LARGE_SET_EXTRA = {
    "scircuit": (170_998, 170_998, 958_936),
    "af_shell10": (1_508_065, 1_508_065, 52_672_325),
}


# ------------------------------------------------------------------------------------------------------
# numpy generators (small)
# ------------------------------------------------------------------------------------------------------
def csr_from_row_lengths(lens, n, rng, locality=64, far_fraction=0.1):
    """CSR with the given row lengths; columns near the scaled diagonal plus a fraction of far ones."""
    lens = np.asarray(lens, dtype=np.int64)
    m = lens.size
    rowptr = np.zeros(m + 1, dtype=np.int64)
    np.cumsum(lens, out=rowptr[1:])
    nnz = int(rowptr[-1])
    assert nnz < 2**31 - 1
    rows = np.repeat(np.arange(m, dtype=np.int64), lens)
    base = (rows * n) // max(m, 1)
    near = base + rng.integers(-locality, locality + 1, size=nnz)
    far = rng.integers(0, max(n, 1), size=nnz)
    pick_far = rng.random(nnz) < far_fraction
    cols = np.where(pick_far, far, near)
    cols = np.clip(cols, 0, max(n - 1, 0)).astype(np.int32)
    vals = rng.uniform(-1.0, 1.0, size=nnz)
    return rowptr.astype(np.int32), cols, vals


def random_csr(m, n, avg, seed, kind="uniform"):
    """kinds: uniform | short | powerlaw | spikes | empty_rows | dense_rows | single"""
    rng = np.random.default_rng(seed)
    if kind == "uniform":
        lens = rng.integers(max(avg - 2, 0), avg + 3, size=m)
    elif kind == "short":
        lens = rng.integers(0, 4, size=m)
    elif kind == "powerlaw":
        lens = np.minimum((rng.pareto(1.1, size=m) * avg).astype(np.int64), max(n, 1) * 4)
    elif kind == "spikes":
        lens = rng.integers(0, 6, size=m)
        k = max(1, m // 200)
        lens[rng.integers(0, m, size=k)] = rng.integers(3000, 20000, size=k)
    elif kind == "empty_rows":
        lens = rng.integers(0, 2 * avg + 1, size=m)
        lens[rng.random(m) < 0.6] = 0
        lens[: m // 10] = 0
        lens[-(m // 10 + 1):] = 0
    elif kind == "dense_rows":
        lens = rng.integers(300, 600, size=m)
    elif kind == "single":
        lens = np.ones(m, dtype=np.int64)
    else:
        raise ValueError(kind)
    return csr_from_row_lengths(lens, n, rng)


def rajat03_like(seed=0xC1):
    """Stand-in for examples/data/rajat03.csr (7602 x 7602, examples/batch.sh:51-52): circuit pattern =
    diagonal + a few off-diagonals per row + a handful of long rows (~4.3 nnz/row)."""
    rng = np.random.default_rng(seed)
    m = n = 7602
    lens = 1 + rng.integers(1, 6, size=m)
    lens[rng.integers(0, m, size=6)] = rng.integers(150, 600, size=6)  # total ~32.6 k nnz like rajat03
    rowptr, cols, vals = csr_from_row_lengths(lens, n, rng, locality=40, far_fraction=0.15)
    cols[rowptr[:-1]] = np.arange(m, dtype=np.int32)  # first entry of every row is the diagonal
    return rowptr, cols, vals


def banded_csr(m, offsets=(-4, -3, -2, -1, 0, 1, 2, 3), first_row=0, total_rows=None):
    """Rows [first_row, first_row + m) of the banded matrix of SURVEY.md 8(d) C5 (global column ids,
    offsets clipped at the matrix edges, value 1/(1+|off|) with an alternating sign)."""
    total = total_rows if total_rows is not None else m
    rows = np.arange(first_row, first_row + m, dtype=np.int64)
    offs = np.asarray(offsets, dtype=np.int64)
    cols = rows[:, None] + offs[None, :]
    ok = (cols >= 0) & (cols < total)
    lens = ok.sum(axis=1)
    rowptr = np.zeros(m + 1, dtype=np.int64)
    np.cumsum(lens, out=rowptr[1:])
    sign = np.where((rows[:, None] + offs[None, :]) % 2 == 0, 1.0, -1.0)
    vals = sign / (1.0 + np.abs(offs)[None, :])
    return rowptr.astype(np.int32), cols[ok].astype(np.int32), vals[ok].astype(np.float64)


def reference_rand_grid(n, rng):
    """Vectors on the reference's 100-point grid: -1 + 2*(k % 100)/101 (cli/utils.hpp:46-49), with k
    from a seeded numpy generator instead of libc rand()."""
    return -1.0 + 2.0 * (rng.integers(0, 2**31 - 1, size=n) % 100) / 101.0


# ------------------------------------------------------------------------------------------------------
# torch generators (BASELINE.json sizes, built on the GPU)
# ------------------------------------------------------------------------------------------------------
def _exact_lengths_torch(m, nnz, gen, device):
    """Row lengths floor(avg) + Bernoulli(frac) + a mean-preserving +-1 jitter, nudged to sum to nnz."""
    import torch

    avg = nnz / m
    lo = int(np.floor(avg))
    frac = avg - lo
    u = torch.rand(m, generator=gen, device=device)
    j = torch.rand(m, generator=gen, device=device)
    lens = torch.full((m,), lo, dtype=torch.int64, device=device) + (u < frac).to(torch.int64)
    if lo >= 2:
        lens += (j < 0.1).to(torch.int64) - (j > 0.9).to(torch.int64)
    diff = nnz - int(lens.sum().item())
    if diff != 0:  # spread the correction over |diff| evenly spaced rows (+-1 each)
        k = abs(diff)
        assert k <= m, "row-length correction larger than the row count"
        idx = torch.linspace(0, m - 1, k, device=device).to(torch.int64)
        if diff > 0:
            lens[idx] += 1
        else:
            ok = lens[idx] > 0
            lens[idx[ok]] -= 1
            rest = nnz - int(lens.sum().item())  # rows that were already empty could not give one back
            if rest != 0:
                big = torch.argsort(lens, descending=True)[: abs(rest)]
                lens[big] += 1 if rest > 0 else -1
    assert int(lens.sum().item()) == nnz
    return lens


def structured_csr_torch(m, n, nnz, seed, device="cuda", far_fraction=0.10, spread=None):
    """FEM / circuit-like stand-in with exactly (m, n, nnz): row lengths concentrated around nnz/m,
    columns in a strictly increasing run around the scaled diagonal, `far_fraction` of them replaced by
    uniformly random columns.  Returns int32 rowptr, int32 colindex, fp64 values on `device`."""
    import torch

    gen = torch.Generator(device=device)
    gen.manual_seed(seed)
    lens = _exact_lengths_torch(m, nnz, gen, device)
    rowptr64 = torch.zeros(m + 1, dtype=torch.int64, device=device)
    torch.cumsum(lens, 0, out=rowptr64[1:])
    assert int(rowptr64[-1].item()) == nnz
    rows = torch.repeat_interleave(torch.arange(m, device=device, dtype=torch.int64), lens, output_size=nnz)
    k = torch.arange(nnz, device=device, dtype=torch.int64) - rowptr64[rows]
    stride_choices = torch.tensor([1, 1, 2, 3] if spread is None else spread, device=device, dtype=torch.int64)
    stride = stride_choices[torch.randint(0, stride_choices.numel(), (m,), generator=gen, device=device)][rows]
    base = (rows * n) // m - (lens[rows] * stride) // 2
    cols = base + k * stride
    far = torch.rand(nnz, generator=gen, device=device) < far_fraction
    cols = torch.where(far, torch.randint(0, n, (nnz,), generator=gen, device=device, dtype=torch.int64), cols)
    cols.clamp_(0, n - 1)
    vals = torch.rand(nnz, generator=gen, device=device, dtype=torch.float64) * 2.0 - 1.0
    return rowptr64.to(torch.int32), cols.to(torch.int32), vals


def hardesty3_like_torch(device="cuda", seed=0xC2, scale=1.0):
    """Stand-in for SuiteSparse Hardesty3 (8,217,820 x 7,591,564, 40,451,632 nnz, 4.92 nnz/row --
    examples/large-data-set-batch.sh:39-40).  `scale` < 1 shrinks rows/cols/nnz proportionally."""
    m0, n0, nnz0 = LARGE_SET["Hardesty3"]
    m, n = max(int(m0 * scale), 1), max(int(n0 * scale), 1)
    nnz = int(round(nnz0 * (m / m0)))
    return (m, n, nnz) + structured_csr_torch(m, n, nnz, seed, device=device)


def large_set_like_torch(name, device="cuda", seed=0xC300, scale=1.0):
    m0, n0, nnz0 = LARGE_SET[name] if name in LARGE_SET else LARGE_SET_EXTRA[name]
    m, n = max(int(m0 * scale), 1), max(int(n0 * scale), 1)
    nnz = int(round(nnz0 * (m / m0)))
    spread = [1, 1, 1, 2]  # FEM blocks: mostly contiguous column runs
    return (m, n, nnz) + structured_csr_torch(m, n, nnz, seed, device=device, far_fraction=0.02, spread=spread)


SWEEP_NAMES = tuple(LARGE_SET) + tuple(LARGE_SET_EXTRA)  # BASELINE configs[2]: the 12 stand-ins, in sweep order


def sweep_standin_torch(name, device="cuda"):
    """The stand-in of BASELINE configs[2] called `name`, with the seed every consumer (tests, bench.py's `sweep` leg,
    tools/sweep.py) uses: Hardesty3 is configs[1]'s matrix (10 % far columns), the others FEM-like with 2 %."""
    i = SWEEP_NAMES.index(name)
    if name == "Hardesty3":
        return hardesty3_like_torch(device=device)
    if name in LARGE_SET_EXTRA:
        m, n, nnz = LARGE_SET_EXTRA[name]
        return (m, n, nnz) + structured_csr_torch(m, n, nnz, 0xC30A + i, device=device, far_fraction=0.02, spread=[1, 1, 1, 2])
    return large_set_like_torch(name, device=device, seed=0xC300 + i)


def banded_torch(m, first_row=0, total_rows=None, device="cuda", offsets=(-4, -3, -2, -1, 0, 1, 2, 3)):
    """Shard rows [first_row, first_row+m) of the C5 banded matrix, built on the GPU."""
    import torch

    total = total_rows if total_rows is not None else m
    rows = torch.arange(first_row, first_row + m, device=device, dtype=torch.int64)
    offs = torch.tensor(offsets, device=device, dtype=torch.int64)
    cols = rows[:, None] + offs[None, :]
    ok = (cols >= 0) & (cols < total)
    lens = ok.sum(dim=1)
    rowptr = torch.zeros(m + 1, dtype=torch.int64, device=device)
    torch.cumsum(lens, 0, out=rowptr[1:])
    sign = torch.where(cols % 2 == 0, 1.0, -1.0).to(torch.float64)
    vals = sign / (1.0 + offs.abs().to(torch.float64))[None, :]
    return rowptr.to(torch.int32), cols[ok].to(torch.int32), vals[ok].contiguous()


def banded_interior_torch(m, first_row, device="cuda", offsets=(-4, -3, -2, -1, 0, 1, 2, 3), chunk_rows=1 << 24):
    """Interior rows [first_row, first_row+m) of the same banded matrix (no clipping: every row has len(offsets) entries,
    first_row + offsets[0] >= 0 assumed), written chunk by chunk into preallocated int32 / fp64 arrays so that a shard
    at the int32 limit of nnz (2^31 - 65536) needs no multi-GB int64 temporaries."""
    import torch

    k = len(offsets)
    assert first_row + min(offsets) >= 0 and m * k < 2**31
    offs = torch.tensor(offsets, device=device, dtype=torch.int64)
    rowptr = (torch.arange(m + 1, device=device, dtype=torch.int64) * k).to(torch.int32)
    ci = torch.empty(m * k, dtype=torch.int32, device=device)
    v = torch.empty(m * k, dtype=torch.float64, device=device)
    mag = 1.0 / (1.0 + offs.abs().to(torch.float64))
    for r0 in range(0, m, chunk_rows):
        r1 = min(m, r0 + chunk_rows)
        cols = torch.arange(first_row + r0, first_row + r1, device=device, dtype=torch.int64)[:, None] + offs[None, :]
        ci[r0 * k: r1 * k] = cols.reshape(-1).to(torch.int32)
        v[r0 * k: r1 * k] = (torch.where(cols % 2 == 0, 1.0, -1.0).to(torch.float64) * mag[None, :]).reshape(-1)
        del cols
    return rowptr, ci, v


def rmat_torch(scale, edge_factor=16, abcd=(0.57, 0.19, 0.19, 0.05), seed=0xC4, device="cuda", chunk=1 << 26):
    """R-MAT: 2^scale rows, edge_factor * 2^scale generated edges, duplicates merged."""
    import torch

    gen = torch.Generator(device=device)
    gen.manual_seed(seed)
    m = 1 << scale
    edges = edge_factor * m
    a, b, c, _ = abcd
    keys = []
    for start in range(0, edges, chunk):
        cnt = min(chunk, edges - start)
        r = torch.zeros(cnt, dtype=torch.int64, device=device)
        cc = torch.zeros(cnt, dtype=torch.int64, device=device)
        for _bit in range(scale):
            u = torch.rand(cnt, generator=gen, device=device)
            down = u >= (a + b)  # quadrants c, d: row bit set
            right = ((u >= a) & (u < a + b)) | (u >= a + b + c)  # quadrants b, d: col bit set
            r = (r << 1) | down.to(torch.int64)
            cc = (cc << 1) | right.to(torch.int64)
        keys.append((r << scale) | cc)
    key = torch.cat(keys)
    del keys
    key = torch.unique(key)  # sorted, duplicates merged
    nnz = key.numel()
    rows = key >> scale
    cols = (key & (m - 1)).to(torch.int32)
    del key
    counts = torch.bincount(rows, minlength=m)
    del rows
    rowptr = torch.zeros(m + 1, dtype=torch.int64, device=device)
    torch.cumsum(counts, 0, out=rowptr[1:])
    vals = torch.rand(nnz, generator=gen, device=device, dtype=torch.float64) * 2.0 - 1.0
    return m, m, nnz, rowptr.to(torch.int32), cols, vals


def algorithmic_bytes(m, n, nnz, beta_nonzero=True):
    """Canonical bytes of one SpMV: values + colindex + rowptr + x once + y read & write."""
    return 12 * nnz + 4 * (m + 1) + 8 * n + (16 if beta_nonzero else 8) * m


def reference_bytes(m, nnz):
    """The reference harness' byte count (benchmark/utils/statistics_logger.cpp:43; omits x)."""
    return 8 * (2 * m + nnz) + 4 * (m + 1 + nnz)
