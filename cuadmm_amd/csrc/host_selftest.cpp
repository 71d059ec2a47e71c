// Host-side self test for the sanitizer build (cuadmm_amd/build.py: build_host_sanitized): drives the loader, the block
// bookkeeping, the A*A^T ordering / factorisation / solves (whole, split, threaded) and the schedule model on a problem
// directory, so that AddressSanitizer / UBSan see the code paths the engine uses.  Not part of libcuadmm_amd.so.
#include <cmath>
#include <cstdio>
#include <cstring>
#include <numeric>
#include <string>
#include <vector>

#include "common.h"
#include "cuadmm_amd.h"
#include "sign_sched.h"

using namespace cuadmm;

#define ST_CHECK(cond) do { if (!(cond)) { fprintf(stderr, "host_selftest: check failed at line %d: %s\n", __LINE__, #cond); return __LINE__; } } while (0)

extern "C" int cuadmm_host_selftest(const char* problem_dir, const char* scratch_dir) {
  // --- loader
  ProblemData p;
  ST_CHECK(load_problem_txt(problem_dir, p, false) == CUADMM_OK);
  ST_CHECK(p.mat_num > 0 && p.vec_len > 0 && p.con_num > 0);
  ProblemData bad;
  ST_CHECK(load_problem_txt(std::string(scratch_dir) + "/does_not_exist/", bad, false) == CUADMM_ERR_IO);
  {
    const std::string d = std::string(scratch_dir) + "/";
    FILE* f = fopen((d + "blk.txt").c_str(), "w"); ST_CHECK(f); fputs("s 2\nu 3\n  \ngarbage\n-\n 2 \n", f); fclose(f);
    std::vector<char> t; std::vector<int> sz;
    ST_CHECK(read_blk_file(d + "blk.txt", t, sz) == CUADMM_OK && sz.size() == 3 && t[1] == 'u');
    f = fopen((d + "trip.txt").c_str(), "w"); ST_CHECK(f); fputs("0 0 1.5\n3 1 -2e-3\n\n7 0 4\n", f); fclose(f);
    std::vector<int> r, c; std::vector<double> v;
    ST_CHECK(read_triplets(d + "trip.txt", r, c, v, false) == CUADMM_OK && v.size() == 3);
    std::vector<int> cp;
    coo_to_csc(cp, c, r, v, 3, 2);
    ST_CHECK(cp.size() == 3 && cp[2] == 3);
  }
  // --- block bookkeeping
  std::vector<int> psd;
  for (int b : p.blk) if (b > 0) psd.push_back(b);
  std::vector<int> sizes, nums;
  analyze_blk(psd.data(), (int)psd.size(), sizes, nums);
  MatrixSizes ms; ms.init(sizes, nums);
  long long L = 0;
  for (int b : p.blk) L += blk_svec_len(b);
  ST_CHECK(L == p.vec_len);
  if ((int)psd.size() == p.mat_num) {
    std::vector<int> mB((size_t)L), m1((size_t)L), m2((size_t)L);
    ST_CHECK(cuadmm_get_maps(p.blk.data(), p.mat_num, p.vec_len, mB.data(), m1.data(), m2.data()) == CUADMM_OK);
  }
  for (int world : {1, 2, 3, 8}) {
    std::vector<int> first;
    partition_blocks(p.blk.data(), p.mat_num, world, first);
    ST_CHECK((int)first.size() == world + 1 && first[0] == 0 && first[world] == p.mat_num);
  }
  // --- A (CSC of A = CSR of At): rows of At are columns here
  std::vector<int> Acp((size_t)p.vec_len + 1, 0), Ari((size_t)p.At_vals.size());
  std::vector<double> Ax(p.At_vals.size());
  for (int j = 0; j < p.con_num; ++j) for (int q = p.At_col_ptrs[j]; q < p.At_col_ptrs[j + 1]; ++q) Acp[(size_t)p.At_row_ids[q] + 1]++;
  for (int i = 0; i < p.vec_len; ++i) Acp[(size_t)i + 1] += Acp[i];
  {
    std::vector<int> fill(Acp.begin(), Acp.end() - 1);
    for (int j = 0; j < p.con_num; ++j)
      for (int q = p.At_col_ptrs[j]; q < p.At_col_ptrs[j + 1]; ++q) { const int pos = fill[p.At_row_ids[q]]++; Ari[pos] = j; Ax[pos] = p.At_vals[q]; }
  }
  const int m = p.con_num;
  cuadmm_aat* f = nullptr;
  ST_CHECK(cuadmm_aat_create(m, p.vec_len, Acp.data(), Ari.data(), Ax.data(), 1e-15, &f) == CUADMM_OK);
  const int* perm = cuadmm_aat_perm(f);
  std::vector<char> seen((size_t)m, 0);
  for (int i = 0; i < m; ++i) { ST_CHECK(perm[i] >= 0 && perm[i] < m && !seen[perm[i]]); seen[perm[i]] = 1; }
  std::vector<double> rhs((size_t)m), x((size_t)m);
  for (int i = 0; i < m; ++i) rhs[i] = std::sin(0.37 * i) + 0.5;
  ST_CHECK(cuadmm_aat_solve_permuted(f, rhs.data(), x.data()) == CUADMM_OK);
  for (int i = 0; i < m; ++i) ST_CHECK(std::isfinite(x[i]));
  // split factor: leading sweeps + dense tail extraction
  cuadmm_aat* g = nullptr;
  const int k = m >= 8 ? m / 4 : 0;
  if (k > 0) {
    ST_CHECK(cuadmm_aat_create_split(m, p.vec_len, Acp.data(), Ari.data(), Ax.data(), 1e-15, -k, &g) == CUADMM_OK);
    ST_CHECK(cuadmm_aat_tail_k(g) == k);
    const int64_t* srp; const int* sci; const double* sv;
    ST_CHECK(cuadmm_aat_tail_schur(g, &srp, &sci, &sv) == CUADMM_OK && srp[k] >= k);
    std::vector<double> y(rhs);
    ST_CHECK(cuadmm_aat_solve_leading_forward(g, k, y.data()) == CUADMM_OK);
    ST_CHECK(cuadmm_aat_solve_leading_backward(g, k, y.data()) == CUADMM_OK);
    cuadmm_aat_tail_schur_release(g);
    cuadmm_aat_free(g);
  }
  cuadmm_aat_free(f);
  // --- host pool
  std::vector<double> acc(64, 0.0);
  struct Ctx { double* a; } ctx{acc.data()};
  for (int rep = 0; rep < 50; ++rep)
    cuadmm_host_parallel_for(64, [](int c, void* q) { static_cast<Ctx*>(q)->a[c] += c; }, &ctx);
  for (int c = 0; c < 64; ++c) ST_CHECK(acc[c] == 50.0 * c);
  // --- schedule model
  std::vector<double> s(40);
  for (int i = 0; i < 40; ++i) s[i] = i < 10 ? 0.0 : std::pow(10.0, -0.3 * (i - 10)) * 0.7;
  double err = 0;
  const int steps = cuadmm_sign_sched_simulate(s.data(), 40, 0, &err);
  ST_CHECK(steps > 0 && steps <= SignSched::kCap && err <= 2.5e-13);
  return 0;
}
