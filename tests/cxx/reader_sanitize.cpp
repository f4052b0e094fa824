// Host-only driver for the matrix readers (spmv_acc_amd/cli/matrix_io.hpp), built by tests/test_cli_io.py with
// g++ -fsanitize=address,undefined: reads every file given on the command line with the reader its suffix selects and
// prints one line per file, "ok rows cols nnz" or "error <message>".  A malformed file must end in "error", never in a
// sanitizer report, a crash or an out-of-range structure handed on to the GPU.
#include <iostream>
#include <string>

#include "matrix_io.hpp"

static bool ends_with(const std::string &s, const std::string &suf) {
  return s.size() >= suf.size() && s.compare(s.size() - suf.size(), suf.size(), suf) == 0;
}

int main(int argc, char **argv) {
  for (int i = 1; i < argc; ++i) {
    const std::string path = argv[i];
    try {
      spmv_cli::HostCsr A;
      if (ends_with(path, ".mtx"))
        A = spmv_cli::read_matrix_market(path);
      else if (ends_with(path, ".bin2"))
        A = spmv_cli::read_bin2(path);
      else
        A = spmv_cli::read_csr_text(path);
      spmv_cli::validate_csr(path, A);
      long long checksum = 0;
      for (int r = 0; r < A.rows; ++r)
        for (int j = A.rowptr[r]; j < A.rowptr[r + 1]; ++j) checksum += A.colidx[j];
      std::cout << "ok " << A.rows << " " << A.cols << " " << A.nnz << " " << checksum << std::endl;
    } catch (const std::exception &e) {
      std::cout << "error " << e.what() << std::endl;
    }
  }
  return 0;
}
