// Block bookkeeping: size census, the reference's large/small classes and svec<->dense maps,
// and the contiguous block partition used to shard blocks over GPUs.
//
// svec index contract (reference src/utils/get_maps.cu:116-130): block k (blk.txt order) owns
// svec slots [off_k, off_k + n(n+1)/2); inside a block the slot order is "for i=1..n, for j=1..i",
// i.e. the upper triangle column by column (column i, rows 1..i); off-diagonals carry sqrt(2).
#include <algorithm>
#include <iomanip>
#include <iostream>
#include <set>

#include "common.h"
#include "sign_sched.h"

namespace cuadmm {

// src/matrix_sizes.cu:14-19
bool is_large_mat(int mat_size, int mat_num) {
  if (mat_size > 32) return true;
  return ((double)mat_size - 17.0 > (double)mat_num * 1.4);
}

// src/utils/analyze_blk.cu:63-99 (ascending unique sizes with multiplicities)
void analyze_blk(const int* blk, int mat_num, std::vector<int>& sizes, std::vector<int>& nums) {
  std::set<int> ss(blk, blk + mat_num);
  sizes.assign(ss.begin(), ss.end());
  nums.assign(sizes.size(), 0);
  for (int i = 0; i < mat_num; ++i) {
    size_t j = std::lower_bound(sizes.begin(), sizes.end(), blk[i]) - sizes.begin();
    nums[j]++;
  }
}

void print_blk_census(const std::vector<int>& sizes, const std::vector<int>& nums) {
  std::cout << "\nAnalysis of the blk vector:" << std::endl;
  for (size_t i = 0; i < sizes.size(); ++i) {
    std::cout << "     " << std::setw(4) << nums[i] << " matrices of size " << std::setw(3) << sizes[i];
    std::cout << (is_large_mat(sizes[i], nums[i]) ? " (large)" : " (small)") << std::endl;
  }
}

// src/matrix_sizes.cu:22-69
void MatrixSizes::init(const std::vector<int>& sizes, const std::vector<int>& nums) {
  large_start.assign(1, 0); large_W_start.assign(1, 0);
  small_start.assign(1, 0); small_W_start.assign(1, 0);
  for (size_t i = 0; i < sizes.size(); ++i) {
    int s = sizes[i], c = nums[i];
    bool big = is_large_mat(s, c);
    cls.push_back({s, big});
    if (big) {
      large_num += c; sum_large += (long long)s * c; total_large += (long long)c * s * s;
      large_sizes.push_back(s); large_nums.push_back(c);
      large_start.push_back(total_large); large_W_start.push_back(sum_large);
    } else {
      small_num += c; sum_small += (long long)s * c; total_small += (long long)c * s * s;
      small_sizes.push_back(s); small_nums.push_back(c);
      small_start.push_back(total_small); small_W_start.push_back(sum_small);
    }
  }
}

bool MatrixSizes::is_large(int s) const {
  for (auto& p : cls) if (p.first == s) return p.second;
  return s > 32;
}

// src/matrix_sizes.cu:75-113 (console census, byte-compatible with the reference log)
void MatrixSizes::print() const {
  auto dump = [](const char* what, const std::vector<int>& sz, const std::vector<int>& nm, long long total,
                 long long sum, int num, const std::vector<long long>& start) {
    std::cout << "\nAnalysis of the " << what << " matrices sizes:" << std::endl;
    std::cout << "    size of " << what << " matrices: ";
    for (int v : sz) std::cout << std::setw(3) << v << " ";
    std::cout << std::endl;
    std::cout << "  number of " << what << " matrices: ";
    for (int v : nm) std::cout << std::setw(3) << v << " ";
    std::cout << std::endl;
    std::cout << "    total size of " << what << " matrices: " << total << std::endl;
    std::cout << "  sum of sizes of " << what << " matrices: " << sum << std::endl;
    std::cout << "    nb " << what << " (with multiplicity): " << num << std::endl;
    std::cout << "  " << what << " matrices start indices: ";
    for (long long v : start) std::cout << v << " ";
    std::cout << std::endl;
  };
  dump("large", large_sizes, large_nums, total_large, sum_large, large_num, large_start);
  dump("small", small_sizes, small_nums, total_small, sum_small, small_num, small_start);
}

// Contiguous block ranges per rank, balanced by the projection cost sum n^3.
void partition_blocks(const int* blk, int mat_num, int world, std::vector<int>& first) {
  first.assign((size_t)world + 1, 0);
  auto cost = [](int b) { return b > 0 ? (double)b * b * b : (double)(-(long long)b); };   // 'u' blocks: a copy
  double total = 0;
  for (int i = 0; i < mat_num; ++i) total += cost(blk[i]);
  double acc = 0;
  int r = 1;
  for (int i = 0; i < mat_num && r < world; ++i) {
    acc += cost(blk[i]);
    // close rank r-1 once it holds its share; keep at least the blocks needed by later ranks possible
    while (r < world && acc >= total * r / world) first[r++] = i + 1;
  }
  for (; r < world; ++r) first[r] = mat_num;
  first[world] = mat_num;
}

}  // namespace cuadmm

using namespace cuadmm;

extern "C" {

int cuadmm_is_large_mat(int mat_size, int mat_num) { return is_large_mat(mat_size, mat_num) ? 1 : 0; }

int cuadmm_analyze_blk(const int* blk, int mat_num, int* sizes_out, int* nums_out, int cap) {
  if (!blk || mat_num < 0) { set_error("analyze_blk: bad arguments"); return CUADMM_ERR_INVALID; }
  std::vector<int> s, c;
  analyze_blk(blk, mat_num, s, c);
  for (size_t i = 0; i < s.size() && (int)i < cap; ++i) { sizes_out[i] = s[i]; nums_out[i] = c[i]; }
  return (int)s.size();
}

// src/utils/get_maps.cu:80-135
int cuadmm_get_maps(const int* blk, int mat_num, int vec_len, int* map_B, int* map_M1, int* map_M2) {
  if (!blk || !map_B || !map_M1 || !map_M2) { set_error("get_maps: null argument"); return CUADMM_ERR_INVALID; }
  std::vector<int> sizes, nums;
  analyze_blk(blk, mat_num, sizes, nums);
  MatrixSizes ms;
  ms.init(sizes, nums);
  std::vector<int> seen_large(ms.large_sizes.size(), 0), seen_small(ms.small_sizes.size(), 0);
  long long idx = 0;
  for (int k = 0; k < mat_num; ++k) {
    int s = blk[k];
    bool big = ms.is_large(s);
    long long base;
    if (big) {
      size_t c = std::find(ms.large_sizes.begin(), ms.large_sizes.end(), s) - ms.large_sizes.begin();
      base = ms.large_start[c] + (long long)seen_large[c]++ * s * s;
    } else {
      size_t c = std::find(ms.small_sizes.begin(), ms.small_sizes.end(), s) - ms.small_sizes.begin();
      base = ms.small_start[c] + (long long)seen_small[c]++ * s * s;
    }
    for (int i = 0; i < s; ++i)
      for (int j = 0; j <= i; ++j) {
        if (idx >= vec_len) { set_error("get_maps: vec_len too small"); return CUADMM_ERR_INVALID; }
        map_B[idx] = big ? 0 : 1;
        map_M1[idx] = (int)(base + (long long)s * i + j);
        map_M2[idx] = (int)(base + (long long)s * j + i);
        ++idx;
      }
  }
  if (idx != vec_len) { set_error("get_maps: vec_len %d != sum n(n+1)/2 = %lld", vec_len, idx); return CUADMM_ERR_INVALID; }
  return CUADMM_OK;
}

// src/utils/get_maps.cu:21-68
int cuadmm_get_maps_duo(const int* blk, int mat_num, int LARGE, int SMALL, int vec_len, int* map_B, int* map_M1,
                        int* map_M2) {
  (void)SMALL;
  long long idx = 0;
  int k_mom = 0, k_loc = 0;
  for (int k = 0; k < mat_num; ++k) {
    int s = blk[k];
    int b;
    long long base;
    if (s == LARGE) { b = 0; base = (long long)s * s * k_mom++; }
    else { b = 1; base = (long long)s * s * k_loc++; }
    for (int i = 0; i < s; ++i)
      for (int j = 0; j <= i; ++j) {
        if (idx >= vec_len) { set_error("get_maps_duo: vec_len too small"); return CUADMM_ERR_INVALID; }
        map_B[idx] = b;
        map_M1[idx] = (int)(base + (long long)s * i + j);
        map_M2[idx] = (int)(base + (long long)s * j + i);
        ++idx;
      }
  }
  return CUADMM_OK;
}

// src/utils/inverse_permutation.cu:17-30
int cuadmm_inverse_permutation(const int* perm, int n, int* perm_inv) {
  for (int i = 0; i < n; ++i) {
    if (perm[i] < 0 || perm[i] >= n) { set_error("inverse_permutation: perm[%d]=%d out of range", i, perm[i]); return CUADMM_ERR_INVALID; }
    perm_inv[perm[i]] = i;
  }
  return CUADMM_OK;
}

int cuadmm_partition_blocks(const int* blk, int mat_num, int world, int* first_block_out) {
  if (!blk || world < 1 || !first_block_out) { set_error("partition_blocks: bad arguments"); return CUADMM_ERR_INVALID; }
  std::vector<int> f;
  partition_blocks(blk, mat_num, world, f);
  for (int r = 0; r <= world; ++r) first_block_out[r] = f[r];
  return CUADMM_OK;
}

// row ranges of the dense tail's shards (tail_shard_bound, common.h): out[p] .. out[p + 1] for rank p, world + 1 entries, rows of the
// PADDED triangle (out[world] = the padded size K of a tail of k columns)
int cuadmm_tail_shard_bounds(int k, int world, int* out) {
  if (k < 1 || world < 1 || !out) { set_error("tail_shard_bounds: bad arguments"); return CUADMM_ERR_INVALID; }
  const int K = tail_padded(k);
  for (int p = 0; p <= world; ++p) out[p] = tail_shard_bound(K, p, world);
  return CUADMM_OK;
}

// host model of the adaptive matrix-sign schedule (sign_sched.h): the kernels run the same state machine per block
int cuadmm_sign_sched_simulate(double* s, int n, int lagged, double* max_err_out) {
  if (!s || n < 0) { set_error("sign_sched_simulate: bad arguments"); return CUADMM_ERR_INVALID; }
  return sign_sched_simulate(s, n, lagged, max_err_out);
}

// the same with the schedule's warm start: lift0 = the hint (lift steps of the previous projection), *lifts_out = the next hint
int cuadmm_sign_sched_simulate_hint(double* s, int n, int lagged, int lift0, double* max_err_out, int* lifts_out) {
  if (!s || n < 0) { set_error("sign_sched_simulate: bad arguments"); return CUADMM_ERR_INVALID; }
  return sign_sched_simulate(s, n, lagged, max_err_out, lift0, lifts_out);
}

}  // extern "C"
