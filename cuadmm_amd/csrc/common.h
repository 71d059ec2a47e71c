// Shared host-side declarations of the engine (error plumbing, small helpers).
#pragma once
#include <cmath>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>

#include "cuadmm_amd.h"

namespace cuadmm {

// thread-local message returned by cuadmm_last_error()
void set_error(const char* fmt, ...);
const char* get_error();

struct ProblemData {
  int vec_len = 0, con_num = 0, mat_num = 0;
  std::vector<int> At_col_ptrs, At_row_ids, At_coo_col_ids;
  std::vector<double> At_vals;
  std::vector<int> b_idx, C_idx;
  std::vector<double> b_vals, C_vals;
  std::vector<int> blk;
  std::vector<char> blk_types;
};

// svec slots of a block: n (n + 1) / 2 for a PSD block of size n; an UNCONSTRAINED block of n variables (blk.txt type
// 'u', README.md:55-64 of the reference: "WIP") is carried as the negative size -n through every blk array and owns n slots
inline long long blk_svec_len(int b) { return b >= 0 ? (long long)b * (b + 1) / 2 : -(long long)b; }

// io.cpp
int read_blk_file(const std::string& fn, std::vector<char>& types, std::vector<int>& sizes);
int read_triplets(const std::string& fn, std::vector<int>& r, std::vector<int>& c, std::vector<double>& v, bool allow_missing);
int read_numbers(const std::string& fn, std::vector<double>& vals);
void coo_to_csc(std::vector<int>& col_ptrs, std::vector<int>& col_ids, std::vector<int>& row_ids,
                std::vector<double>& vals, int nnz, int col_num);
int load_problem_txt(const std::string& prefix, ProblemData& p, bool verbose);

// blocks.cpp
bool is_large_mat(int mat_size, int mat_num);
void analyze_blk(const int* blk, int mat_num, std::vector<int>& sizes, std::vector<int>& nums);
struct MatrixSizes {
  std::vector<int> large_sizes, large_nums, small_sizes, small_nums;
  std::vector<long long> large_start, large_W_start, small_start, small_W_start;
  long long total_large = 0, total_small = 0, sum_large = 0, sum_small = 0;
  int large_num = 0, small_num = 0;
  void init(const std::vector<int>& sizes, const std::vector<int>& nums);
  bool is_large(int s) const;
  void print() const;
  std::vector<std::pair<int, bool>> cls;  // (size, is_large)
};
void print_blk_census(const std::vector<int>& sizes, const std::vector<int>& nums);
void partition_blocks(const int* blk, int mat_num, int world, std::vector<int>& first);

// Row boundaries of the tail's shards: rank p of `world` applies rows [tail_shard_bound(K, p, world), tail_shard_bound(K, p + 1, world))
// of W = inv(L22), in the kernels' row numbering (row 0 is the LONGEST row of the triangle: K entries, row r has K - r): equal shares of
// the triangle's ENTRIES -- of the bytes a solve reads -- not of its rows; multiples of 8; [0, K) covered, monotone, and a rank whose
// share rounds away gets an EMPTY range (K = 64, world = 8: ranks 1 and 2 both start at row 8), for which every kernel writes zeros.
inline int tail_shard_bound(int K, int p, int world) {
  if (p <= 0) return 0;
  if (p >= world) return K;
  const double f = (double)p / (double)world;
  int r = (int)((double)K * (1.0 - std::sqrt(1.0 - f)));
  r = (r + 7) / 8 * 8;
  return r < K ? r : K;
}
inline int tail_padded(int k) { return (k + 63) / 64 * 64; }      // TailSolve::K for a tail of k columns

// dense tree tops of the y-solve (lead_solve.h): both packed triangles of the explicit inverses, on the host while they are built
// and on the device afterwards -- ONE limit for the planner (aat_ldlt.cpp, tops_us) and the build (lead_solve.hip, build_tops)
constexpr long long kLeadTopsMaxBytes = 2000000000ll;

}  // namespace cuadmm

// host thread pool of aat_ldlt.cpp (internal): fn(chunk, ctx) for chunk = 0..nchunks-1
extern "C" void cuadmm_host_parallel_for(int nchunks, void (*fn)(int, void*), void* ctx);
extern "C" void cuadmm_host_pool_hint(int ranks_on_node);
extern "C" int cuadmm_host_pool_threads(void);
