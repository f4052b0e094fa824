// ref_shim.cpp -- C entry point onto the REFERENCE's own row-block analysis.  TEST INFRASTRUCTURE ONLY.
//
// oracle/Makefile compiles /root/reference/src/acc/hip-csr-adaptive-plus/csr_adaptive_plus_analyze.cpp
// (unmodified, where it lies) together with this file into oracle/_ref/libref_analyze.so.  That
// reference translation unit needs nothing but <vector>, its own header and its own config header,
// so it builds with plain g++.  This shim only selects one of the template instances the reference
// explicitly instantiates (csr_adaptive_plus_analyze.cpp:100-134) and copies the vectors out.
#include <cstddef>
#include <vector>

#include "csr_adaptive_plus_analyze.h" // from -I/root/reference/src/acc/hip-csr-adaptive-plus

namespace {
template <int T, int V>
int run(int m, int nnz, int min_nnz, const int *rp, std::vector<int> &bp, std::vector<int> &fbr) {
  return csr_adaptive_plus_analyze_imp<int, T, V>(m, nnz, min_nnz, bp, fbr, rp, rp);
}

template <int T>
int by_vec(int vec, int m, int nnz, int min_nnz, const int *rp, std::vector<int> &bp, std::vector<int> &fbr) {
  switch (vec) {
  case 1: return run<T, 1>(m, nnz, min_nnz, rp, bp, fbr);
  case 2: return run<T, 2>(m, nnz, min_nnz, rp, bp, fbr);
  case 4: return run<T, 4>(m, nnz, min_nnz, rp, bp, fbr);
  case 8: return run<T, 8>(m, nnz, min_nnz, rp, bp, fbr);
  case 16: return run<T, 16>(m, nnz, min_nnz, rp, bp, fbr);
  case 32: return run<T, 32>(m, nnz, min_nnz, rp, bp, fbr);
  case 64: return run<T, 64>(m, nnz, min_nnz, rp, bp, fbr);
  default: return -2;
  }
}
} // namespace

// returns the block count, -1 if bp_cap is too small, -2 for a (threads, vec) pair the reference
// does not instantiate.  first_block_of_row gets m+1 entries (zero-initialised like the
// reference's std::vector::resize, csr_adaptive_plus_spmv.cpp:27).
extern "C" int ref_adaptive_plus_analyze(int threads_per_block, int vec_size, int m, int nnz, int min_nnz_per_block,
                                         const int *host_row_ptr, int *break_points, int bp_cap,
                                         int *first_block_of_row) {
  std::vector<int> bp;
  std::vector<int> fbr(static_cast<std::size_t>(m) + 1);
  int blocks;
  switch (threads_per_block) {
  case 256: blocks = by_vec<256>(vec_size, m, nnz, min_nnz_per_block, host_row_ptr, bp, fbr); break;
  case 512: blocks = by_vec<512>(vec_size, m, nnz, min_nnz_per_block, host_row_ptr, bp, fbr); break;
  case 1024: blocks = by_vec<1024>(vec_size, m, nnz, min_nnz_per_block, host_row_ptr, bp, fbr); break;
  default: return -2;
  }
  if (blocks < 0) return blocks;
  if (static_cast<int>(bp.size()) > bp_cap) return -1;
  for (std::size_t i = 0; i < bp.size(); i++) break_points[i] = bp[i];
  for (std::size_t i = 0; i < fbr.size(); i++) first_block_of_row[i] = fbr[i];
  return blocks;
}
