// matrix_io.hpp -- host-side matrix input for spmv-cli: the three formats the reference's CLI accepts
// (cli/main.cpp:36-40: "-f csr | mtx | bin2").  Fresh implementations of the same on-disk contracts:
//   * .csr text  (cli/csr_mtx_reader.hpp:49-91):   5 lines -- free-form header, values, colindex, rowptr, dense x;
//                 rows = len(rowptr) - 1, cols = len(x), nnz = len(values)
//   * bin2       (cli/csr_binary_reader.hpp:37-101, written by tools/suitesparse-dl/conv/conv.go:120-193):
//                 i32 magic 0x20211015, i32 version 2, i32 valtype {1 pattern, 2 int, 3 real, 4 complex}, i32 rows, cols, nnz,
//                 i32 rowptr[rows+1], i32 colindex[nnz], then values (none / i32 / f64)
//   * MatrixMarket coordinate (cli/matrix_market_reader.hpp:50-302): general / symmetric / Hermitian (off-diagonals
//                 mirrored), pattern / real / integer (/ complex: real part), 1-based; entries sorted by (row, col)
//                 (cli/sparse_format.h:100-128)
// Differences on purpose: integer bin2 values are read as nnz int32 (the reference reads 8*nnz bytes
// into a 4*nnz buffer), integer tokens of a .csr file are parsed as integers, and every failure throws instead of
// silently returning an empty matrix.
#pragma once

#include <algorithm>
#include <cerrno>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <numeric>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

namespace spmv_cli {

struct HostCsr {
  int rows = 0, cols = 0, nnz = 0;
  std::vector<int> rowptr;
  std::vector<int> colidx;
  std::vector<double> values;
  std::vector<double> x; // only the .csr text format carries the dense vector
  int valtype = 3;       // what the file declared: 1 pattern, 2 integer, 3 real, 4 complex (real parts kept) -- the bin2 header's code
};

// Structural checks every reader ends with: a file that fails them would send the kernels out of bounds (the reference's
// readers hand such files straight to the GPU).
inline void validate_csr(const std::string &path, const HostCsr &A) {
  if (A.rows < 0 || A.cols < 0 || A.nnz < 0 || A.rowptr.size() != static_cast<size_t>(A.rows) + 1 ||
      A.colidx.size() != static_cast<size_t>(A.nnz) || A.values.size() != static_cast<size_t>(A.nnz))
    throw std::runtime_error(path + ": array sizes do not match the header");
  if (A.rowptr.front() != 0 || A.rowptr.back() != A.nnz) throw std::runtime_error(path + ": rowptr does not match nnz");
  for (size_t i = 0; i + 1 < A.rowptr.size(); ++i)
    if (A.rowptr[i] > A.rowptr[i + 1]) throw std::runtime_error(path + ": rowptr decreases at row " + std::to_string(i));
  for (size_t j = 0; j < A.colidx.size(); ++j)
    if (A.colidx[j] < 0 || A.colidx[j] >= A.cols)
      throw std::runtime_error(path + ": column index " + std::to_string(A.colidx[j]) + " outside [0, " + std::to_string(A.cols) +
                               ") at non-zero " + std::to_string(j));
}

inline std::string slurp(const std::string &path) {
  std::ifstream f(path, std::ios::in | std::ios::binary);
  if (!f) throw std::runtime_error("cannot open " + path);
  std::ostringstream ss;
  ss << f.rdbuf();
  return ss.str();
}

// ---- .csr text ------------------------------------------------------------------------------------------
namespace detail {
template <typename T, typename Conv> void parse_line(const char *b, const char *e, std::vector<T> &out, Conv conv) {
  const char *p = b;
  while (p < e) {
    while (p < e && (*p == ' ' || *p == '\t' || *p == '\r')) ++p;
    if (p >= e) break;
    char *next = nullptr;
    out.push_back(conv(p, &next));
    if (next == p) throw std::runtime_error("bad token in .csr file");
    p = next;
  }
}
} // namespace detail

inline HostCsr read_csr_text(const std::string &path) {
  const std::string buf = slurp(path);
  // split into the first five lines
  std::vector<std::pair<const char *, const char *>> lines;
  const char *p = buf.data(), *end = buf.data() + buf.size();
  while (p < end && lines.size() < 5) {
    const char *nl = static_cast<const char *>(std::memchr(p, '\n', static_cast<size_t>(end - p)));
    const char *le = nl ? nl : end;
    lines.emplace_back(p, le);
    p = nl ? nl + 1 : end;
  }
  if (lines.size() < 5) throw std::runtime_error(path + ": a .csr file has 5 lines (header, values, colindex, rowptr, x)");
  HostCsr A;
  detail::parse_line(lines[1].first, lines[1].second, A.values, [](const char *s, char **n) { return std::strtod(s, n); });
  detail::parse_line(lines[2].first, lines[2].second, A.colidx,
                     [](const char *s, char **n) { return static_cast<int>(std::strtol(s, n, 10)); });
  detail::parse_line(lines[3].first, lines[3].second, A.rowptr,
                     [](const char *s, char **n) { return static_cast<int>(std::strtol(s, n, 10)); });
  detail::parse_line(lines[4].first, lines[4].second, A.x, [](const char *s, char **n) { return std::strtod(s, n); });
  if (A.rowptr.empty()) throw std::runtime_error(path + ": empty rowptr line");
  A.rows = static_cast<int>(A.rowptr.size()) - 1;
  A.cols = static_cast<int>(A.x.size());
  A.nnz = static_cast<int>(A.values.size());
  if (A.colidx.size() != A.values.size() || A.rowptr.back() != A.nnz || A.rowptr.front() != 0)
    throw std::runtime_error(path + ": inconsistent .csr file (nnz / rowptr mismatch)");
  validate_csr(path, A);
  return A;
}

// ---- bin2 --------------------------------------------------------------------------------------------------
inline HostCsr read_bin2(const std::string &path) {
  std::ifstream f(path, std::ios::in | std::ios::binary);
  if (!f) throw std::runtime_error("cannot open " + path);
  int32_t hdr[6];
  f.read(reinterpret_cast<char *>(hdr), sizeof(hdr));
  if (!f) throw std::runtime_error(path + ": truncated bin2 header");
  if (hdr[0] != 0x20211015) throw std::runtime_error(path + ": bad magic number (not a bin2 file)");
  if (hdr[1] != 2) throw std::runtime_error(path + ": only bin format version 2 is supported");
  const int32_t valtype = hdr[2];
  if (valtype < 1 || valtype > 4) throw std::runtime_error(path + ": unsupported value type");
  HostCsr A;
  A.valtype = valtype;
  A.rows = hdr[3];
  A.cols = hdr[4];
  A.nnz = hdr[5];
  if (A.rows < 0 || A.cols < 0 || A.nnz < 0) throw std::runtime_error(path + ": negative dimension");
  {
    // size the body from the header BEFORE allocating: a corrupt nnz must not reserve gigabytes
    const std::streampos here = f.tellg();
    f.seekg(0, std::ios::end);
    const long long body = static_cast<long long>(f.tellg()) - static_cast<long long>(here);
    f.seekg(here);
    const long long per_value = valtype == 1 ? 0 : (valtype == 2 ? 4 : 8);
    const long long need = 4LL * (static_cast<long long>(A.rows) + 1) + (4LL + per_value) * A.nnz;
    if (body < need) throw std::runtime_error(path + ": truncated bin2 body");
  }
  A.rowptr.resize(static_cast<size_t>(A.rows) + 1);
  A.colidx.resize(static_cast<size_t>(A.nnz));
  A.values.resize(static_cast<size_t>(A.nnz));
  f.read(reinterpret_cast<char *>(A.rowptr.data()), static_cast<std::streamsize>(sizeof(int32_t) * A.rowptr.size()));
  f.read(reinterpret_cast<char *>(A.colidx.data()), static_cast<std::streamsize>(sizeof(int32_t) * A.colidx.size()));
  if (valtype == 1) {
    std::fill(A.values.begin(), A.values.end(), 1.0);
  } else if (valtype == 2) {
    std::vector<int32_t> tmp(static_cast<size_t>(A.nnz));
    f.read(reinterpret_cast<char *>(tmp.data()), static_cast<std::streamsize>(sizeof(int32_t) * tmp.size()));
    std::copy(tmp.begin(), tmp.end(), A.values.begin());
  } else {
    f.read(reinterpret_cast<char *>(A.values.data()), static_cast<std::streamsize>(sizeof(double) * A.values.size()));
  }
  if (!f) throw std::runtime_error(path + ": truncated bin2 body");
  validate_csr(path, A);
  return A;
}

// ---- COO -> CSR --------------------------------------------------------------------------------------------------
struct CooEntry {
  int r, c;
  double v;
};

// entries ordered by (row, column); equal (row, column) pairs keep file order (stable)
inline HostCsr coo_to_csr(int rows, int cols, std::vector<CooEntry> &e) {
  std::stable_sort(e.begin(), e.end(), [](const CooEntry &a, const CooEntry &b) { return a.r != b.r ? a.r < b.r : a.c < b.c; });
  HostCsr A;
  A.rows = rows;
  A.cols = cols;
  A.nnz = static_cast<int>(e.size());
  A.rowptr.assign(static_cast<size_t>(rows) + 1, 0);
  A.colidx.resize(e.size());
  A.values.resize(e.size());
  for (size_t i = 0; i < e.size(); ++i) {
    A.colidx[i] = e[i].c;
    A.values[i] = e[i].v;
    ++A.rowptr[static_cast<size_t>(e[i].r) + 1];
  }
  std::partial_sum(A.rowptr.begin(), A.rowptr.end(), A.rowptr.begin());
  return A;
}

// ---- MatrixMarket ----------------------------------------------------------------------------------------------------
inline HostCsr read_matrix_market(const std::string &path) {
  const std::string buf = slurp(path);
  std::istringstream in(buf);
  std::string line;
  if (!std::getline(in, line)) throw std::runtime_error(path + ": empty file");
  std::istringstream hs(line);
  std::string banner, object, format, field, symmetry;
  hs >> banner >> object >> format >> field >> symmetry;
  auto lower = [](std::string s) {
    for (auto &ch : s) ch = static_cast<char>(::tolower(static_cast<unsigned char>(ch)));
    return s;
  };
  if (banner != "%%MatrixMarket" || lower(object) != "matrix" || lower(format) != "coordinate")
    throw std::runtime_error(path + ": can only read MatrixMarket files in coordinate form");
  field = lower(field);
  symmetry = lower(symmetry);
  const bool pattern = field == "pattern";
  const bool complex_field = field == "complex";
  if (!pattern && !complex_field && field != "real" && field != "integer" && field != "double")
    throw std::runtime_error(path + ": unsupported MatrixMarket field '" + field + "'");
  const bool mirror = symmetry == "symmetric" || symmetry == "hermitian" || symmetry == "skew-symmetric";
  if (!mirror && symmetry != "general") throw std::runtime_error(path + ": unsupported MatrixMarket symmetry '" + symmetry + "'");
  const double mirror_sign = symmetry == "skew-symmetric" ? -1.0 : 1.0;
  long rows = 0, cols = 0, declared = 0;
  while (std::getline(in, line)) {
    if (line.empty() || line[0] == '%') continue;
    std::istringstream ss(line);
    if (!(ss >> rows >> cols >> declared)) throw std::runtime_error(path + ": bad size line");
    break;
  }
  constexpr long kMaxEntries = 2147483647L - 65536; // the library's per-call nnz limit (include/spmv_acc.h)
  if (rows < 0 || cols < 0 || declared < 0 || rows > kMaxEntries || cols > kMaxEntries || declared > kMaxEntries)
    throw std::runtime_error(path + ": size line out of the int32 range");
  std::vector<CooEntry> e;
  // an entry line has at least 4 characters ("1 1\n"): never reserve more than the file can hold
  e.reserve(static_cast<size_t>(std::min<long>(declared, static_cast<long>(buf.size() / 4) + 1)) * (mirror ? 2 : 1));
  long seen = 0;
  while (std::getline(in, line)) {
    const char *s = line.c_str();
    while (*s == ' ' || *s == '\t') ++s;
    if (*s == '\0' || *s == '%' || *s == '\r') continue;
    char *n1 = nullptr, *n2 = nullptr, *n3 = nullptr;
    const long r = std::strtol(s, &n1, 10);
    const long c = std::strtol(n1, &n2, 10);
    if (n1 == s || n2 == n1) throw std::runtime_error(path + ": bad entry line: " + line);
    double v = 1.0;
    if (!pattern) {
      v = std::strtod(n2, &n3);
      if (n3 == n2) throw std::runtime_error(path + ": missing value: " + line);
    }
    if (r < 1 || r > rows || c < 1 || c > cols) throw std::runtime_error(path + ": index out of range: " + line);
    e.push_back({static_cast<int>(r - 1), static_cast<int>(c - 1), v});
    if (mirror && r != c) e.push_back({static_cast<int>(c - 1), static_cast<int>(r - 1), mirror_sign * v});
    ++seen;
  }
  if (seen != declared)
    throw std::runtime_error(path + ": expected " + std::to_string(declared) + " entries, found " + std::to_string(seen));
  if (e.size() > static_cast<size_t>(kMaxEntries)) throw std::runtime_error(path + ": more non-zeros than one call can take");
  HostCsr A = coo_to_csr(static_cast<int>(rows), static_cast<int>(cols), e);
  A.valtype = pattern ? 1 : (field == "integer" ? 2 : (complex_field ? 4 : 3));
  validate_csr(path, A);
  return A;
}

// ---- writer: the bin2 file the reference's converter makes ------------------------------------------------------------
// What `suitesparse-dl conv` writes for a MatrixMarket file (tools/suitesparse-dl/conv/conv.go:92-150, little-endian): magic, version 2, the
// value type of the source, rows, cols, nnz, rowptr, colindex, then the values as the type says -- none for a pattern matrix, int32 for an
// integer one (conv.go:176-186 converts by truncation, as here), f64 otherwise.  The reference's csr_binary_reader.hpp and read_bin2 above
// read it back.  (The reference's tool is Go; this is its one function a user of spmv-cli needs: `spmv-cli in.mtx -f mtx --convert-bin2 out.bin2`.)
inline void write_bin2(const std::string &path, const HostCsr &A) {
  std::ofstream f(path, std::ios::out | std::ios::binary);
  if (!f) throw std::runtime_error("cannot write " + path);
  const int32_t valtype = A.valtype >= 1 && A.valtype <= 4 ? A.valtype : 3;
  const int32_t hdr[6] = {0x20211015, 2, valtype, A.rows, A.cols, A.nnz};
  f.write(reinterpret_cast<const char *>(hdr), sizeof(hdr));
  f.write(reinterpret_cast<const char *>(A.rowptr.data()), static_cast<std::streamsize>(sizeof(int32_t) * A.rowptr.size()));
  f.write(reinterpret_cast<const char *>(A.colidx.data()), static_cast<std::streamsize>(sizeof(int32_t) * A.colidx.size()));
  if (valtype == 2) {
    std::vector<int32_t> iv(A.values.size());
    for (size_t i = 0; i < iv.size(); ++i) iv[i] = static_cast<int32_t>(A.values[i]);
    f.write(reinterpret_cast<const char *>(iv.data()), static_cast<std::streamsize>(sizeof(int32_t) * iv.size()));
  } else if (valtype != 1) {
    f.write(reinterpret_cast<const char *>(A.values.data()), static_cast<std::streamsize>(sizeof(double) * A.values.size()));
  }
  f.flush();
  if (!f) throw std::runtime_error("write failed: " + path);
}

} // namespace spmv_cli
