// Host-side check of the device eigen-solver's control flow (psd_device.h compiled with a one-thread
// group): hipcc -x hip --offload-arch=gfx950 tools/host_check_eig.cpp -I cuadmm_amd/csrc -I include -o /tmp/host_check_eig
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <random>
#include <vector>
#include "psd_device.h"
using namespace cuadmm;
struct HostGroup {
  static constexpr int kSize = 1;
  static constexpr bool kMultiWave = true;
  __host__ __device__ static int rank() { return 0; }
  __host__ __device__ static double sum(double x, double*) { return x; }
  __host__ __device__ static void sync() {}
  template <class Pred>
  __host__ __device__ static int first_true(int l, int n, Pred pred) {
    for (int i = l; i < n; ++i) if (pred(i)) return i;
    return n;
  }
};
int main() {
  std::mt19937_64 rng(1);
  std::normal_distribution<double> nd;
  double worst = 0;
  for (int n : {1, 2, 3, 5, 8, 17, 32, 55, 91, 130}) {
    for (int trial = 0; trial < 3; ++trial) {
      int ld = n | 1;
      std::vector<double> A(n * n), M(n * ld), d(n), e(n), tau(n), vv(n), ww(n), sc(8);
      for (int i = 0; i < n; ++i) for (int j = 0; j <= i; ++j) { double v = nd(rng); if (trial == 1) v = (i == j) ? v : 0; A[i * n + j] = A[j * n + i] = v; }
      if (trial == 2) { std::vector<double> u(n); for (auto& x : u) x = nd(rng); for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) A[i * n + j] = u[i] * u[j]; }
      for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) M[i * ld + j] = A[i * n + j];
      int fail = sym_eig_inplace<HostGroup>(M.data(), ld, n, d.data(), e.data(), tau.data(), vv.data(), ww.data(), d.data(), e.data(), sc.data());
      // residual ||A Z - Z diag(d)|| and orthogonality
      double res = 0, orth = 0, nrm = 0;
      for (int i = 0; i < n; ++i) for (int k = 0; k < n; ++k) {
        double s = 0; for (int j = 0; j < n; ++j) s += A[i * n + j] * M[j * ld + k];
        res = std::fmax(res, std::fabs(s - M[i * ld + k] * d[k])); nrm = std::fmax(nrm, std::fabs(A[i * n + k]));
      }
      for (int a = 0; a < n; ++a) for (int b = 0; b < n; ++b) { double s = 0; for (int r = 0; r < n; ++r) s += M[r * ld + a] * M[r * ld + b]; orth = std::fmax(orth, std::fabs(s - (a == b))); }
      worst = std::fmax(worst, std::fmax(res / std::fmax(nrm, 1.0), orth));
      if (fail || res > 1e-11 * n * std::fmax(nrm, 1.0) || orth > 1e-11 * n) printf("BAD n=%d trial=%d fail=%d res=%g orth=%g\n", n, trial, fail, res, orth);
    }
  }
  printf("host check worst %.3e\n", worst);
  return 0;
}
