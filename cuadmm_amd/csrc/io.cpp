// TXT problem loader: the cuadmm_exe input format (blk.txt, con_num.txt, At.txt, b.txt, C.txt).
//
// Behaviour follows the reference loader (src/utils/io.cu, src/problem.cu:11-83):
//   * At.txt  lines "svec_row constraint_col value", 0-based, any order (sorted on load)
//   * b.txt / C.txt lines "idx 0 value"
//   * blk.txt "s 10" or bare "10"; anything else on a line is ignored
// Parsing is a single pass over the file bytes (At.txt of the larger examples has millions of lines).
#include <algorithm>
#include <cctype>
#include <cerrno>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <numeric>

#include "common.h"

namespace cuadmm {

static thread_local std::string g_err;
void set_error(const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_err = buf;
}
const char* get_error() { return g_err.c_str(); }

static bool slurp(const std::string& fn, std::string& out) {
  std::ifstream f(fn, std::ios::binary);
  if (!f.is_open()) return false;
  f.seekg(0, std::ios::end);
  std::streamoff n = f.tellg();
  f.seekg(0, std::ios::beg);
  out.resize((size_t)n);
  if (n > 0) f.read(&out[0], n);
  return true;
}

int read_numbers(const std::string& fn, std::vector<double>& vals) {
  std::string s;
  if (!slurp(fn, s)) {
    set_error("ERROR: could not open file '%s'. Please verify that the provided directory path is correct.", fn.c_str());
    return CUADMM_ERR_IO;
  }
  vals.clear();
  const char* p = s.c_str();
  const char* end = p + s.size();
  while (p < end) {
    char* q;
    double v = strtod(p, &q);
    if (q == p) break;  // same stop rule as `while (file >> val)`
    vals.push_back(v);
    p = q;
  }
  return CUADMM_OK;
}

// "row col val" triplets (io.cu:96-125; the same grammar serves sparse vectors, io.cu:68-93)
int read_triplets(const std::string& fn, std::vector<int>& r, std::vector<int>& c, std::vector<double>& v,
                  bool allow_missing) {
  std::string s;
  r.clear(); c.clear(); v.clear();
  if (!slurp(fn, s)) {
    if (allow_missing) return CUADMM_OK;
    set_error("Failed to open file: %s", fn.c_str());
    return CUADMM_ERR_IO;
  }
  const char* p = s.c_str();
  while (true) {
    char* q;
    long a = strtol(p, &q, 10);
    if (q == p) break;
    p = q;
    long b = strtol(p, &q, 10);
    if (q == p) break;
    p = q;
    double val = strtod(p, &q);
    if (q == p) break;
    p = q;
    r.push_back((int)a); c.push_back((int)b); v.push_back(val);
  }
  return CUADMM_OK;
}

// read_blk (io.cu:296-329): ^\s*([a-zA-Z])\s+(-?\d+)\s*$  |  ^\s*(-?\d+)\s*$ ; other lines ignored
int read_blk_file(const std::string& fn, std::vector<char>& types, std::vector<int>& sizes) {
  std::ifstream f(fn);
  if (!f.is_open()) {
    set_error("ERROR: could not open file %s", fn.c_str());
    return CUADMM_ERR_IO;
  }
  types.clear(); sizes.clear();
  std::string line;
  while (std::getline(f, line)) {
    const char* p = line.c_str();
    while (*p && isspace((unsigned char)*p)) ++p;
    char type = 's';
    if (isalpha((unsigned char)*p)) {
      type = *p++;
      if (!isspace((unsigned char)*p)) continue;  // letter must be followed by whitespace
      while (*p && isspace((unsigned char)*p)) ++p;
    }
    const char* num = p;
    if (*p == '-') ++p;
    if (!isdigit((unsigned char)*p)) continue;
    while (isdigit((unsigned char)*p)) ++p;
    const char* numend = p;
    while (*p && isspace((unsigned char)*p)) ++p;
    if (*p) continue;  // trailing garbage -> malformed line, ignored
    types.push_back(type);
    sizes.push_back((int)strtol(std::string(num, numend).c_str(), nullptr, 10));
  }
  return CUADMM_OK;
}

// COO -> CSC sorted by (col,row) (io.cu:187-243).  The reference's col_ptrs fill is wrong when
// column 0 is empty; this is the correct counting-sort conversion, which
// coincides on every input where the reference is right.
void coo_to_csc(std::vector<int>& col_ptrs, std::vector<int>& col_ids, std::vector<int>& row_ids,
                std::vector<double>& vals, int nnz, int col_num) {
  col_ptrs.assign((size_t)col_num + 1, 0);
  std::vector<int> order((size_t)nnz);
  std::iota(order.begin(), order.end(), 0);
  std::stable_sort(order.begin(), order.end(), [&](int a, int b) {
    if (col_ids[a] != col_ids[b]) return col_ids[a] < col_ids[b];
    return row_ids[a] < row_ids[b];
  });
  std::vector<int> c2((size_t)nnz), r2((size_t)nnz);
  std::vector<double> v2((size_t)nnz);
  for (int i = 0; i < nnz; ++i) {
    c2[i] = col_ids[order[i]]; r2[i] = row_ids[order[i]]; v2[i] = vals[order[i]];
    if (c2[i] >= 0 && c2[i] < col_num) col_ptrs[(size_t)c2[i] + 1]++;
  }
  for (int j = 0; j < col_num; ++j) col_ptrs[(size_t)j + 1] += col_ptrs[j];
  col_ids.swap(c2); row_ids.swap(r2); vals.swap(v2);
}

// Problem::from_txt (problem.cu:11-83), cold start only (main.cu:14 never passes warm_start)
int load_problem_txt(const std::string& prefix, ProblemData& p, bool verbose) {
  int rc = read_blk_file(prefix + "blk.txt", p.blk_types, p.blk);
  if (rc) return rc;
  p.mat_num = (int)p.blk.size();
  long long L = 0;
  for (int i = 0; i < p.mat_num; ++i) {
    if (p.blk_types[i] == 'u') {  // unconstrained block of n variables (README.md:55-64): carried as -n
      if (p.blk[i] < 1) { set_error("ERROR: block %d 'u %d' in blk.txt", i, p.blk[i]); return CUADMM_ERR_IO; }
      p.blk[i] = -p.blk[i];
    } else if (p.blk_types[i] != 's') {  // problem.cu:28-36
      set_error("ERROR: unknown block type '%c' in blk.txt", p.blk_types[i]);
      return CUADMM_ERR_IO;
    }
    L += blk_svec_len(p.blk[i]);
  }
  if (L > 2147483647LL) { set_error("vector length %lld exceeds int32 (reference API uses int)", L); return CUADMM_ERR_INVALID; }
  p.vec_len = (int)L;
  std::vector<double> cn;
  rc = read_numbers(prefix + "con_num.txt", cn);
  if (rc) return rc;
  if (cn.empty()) { set_error("ERROR: con_num.txt is empty"); return CUADMM_ERR_IO; }
  p.con_num = (int)cn[0];

  rc = read_triplets(prefix + "At.txt", p.At_row_ids, p.At_coo_col_ids, p.At_vals, false);
  if (rc) return rc;
  int nnz = (int)p.At_vals.size();
  for (int i = 0; i < nnz; ++i) {
    if (p.At_row_ids[i] < 0 || p.At_row_ids[i] >= p.vec_len || p.At_coo_col_ids[i] < 0 || p.At_coo_col_ids[i] >= p.con_num) {
      set_error("At.txt entry %d (%d,%d) outside %d x %d", i, p.At_row_ids[i], p.At_coo_col_ids[i], p.vec_len, p.con_num);
      return CUADMM_ERR_IO;
    }
  }
  coo_to_csc(p.At_col_ptrs, p.At_coo_col_ids, p.At_row_ids, p.At_vals, nnz, p.con_num);

  std::vector<int> zero;
  rc = read_triplets(prefix + "b.txt", p.b_idx, zero, p.b_vals, false);
  if (rc) return rc;
  rc = read_triplets(prefix + "C.txt", p.C_idx, zero, p.C_vals, false);
  if (rc) return rc;

  if (verbose) {
    if (nnz > 0) {
      int max_row = *std::max_element(p.At_row_ids.begin(), p.At_row_ids.end());
      if (max_row != p.vec_len - 1)
        std::cerr << "WARNING: the largest column index in At is different from the specified column number!\n" << std::endl;
      int max_col = *std::max_element(p.At_coo_col_ids.begin(), p.At_coo_col_ids.end());
      if (max_col != p.con_num - 1)
        std::cerr << "WARNING: the largest row index in At is different from the SDP vector length!\n" << std::endl;
    }
    std::cout << "Loaded problem from " << prefix << std::endl;
    std::cout << "              vector length: " << p.vec_len << std::endl;
    std::cout << "      number of constraints: " << p.con_num << std::endl;
    std::cout << "           number of blocks: " << p.mat_num << std::endl;
    std::cout << "  number of non-zeros in At: " << nnz << std::endl;
    std::cout << "   number of non-zeros in b: " << p.b_vals.size() << std::endl;
    std::cout << "   number of non-zeros in C: " << p.C_vals.size() << std::endl;
  }
  return CUADMM_OK;
}

}  // namespace cuadmm
