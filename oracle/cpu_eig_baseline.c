/*
 * CPU baseline of the PSD projection: the reference's eig_cpu path.  TEST / BENCH INFRASTRUCTURE ONLY -- nothing under
 * cuadmm_amd/ links or loads this; bench.py's `cpu_baseline` leg and tests/ are the only users.
 *
 * Restates (reference file:line):
 *   - single_eig_lapack                 include/cuadmm/eig_cpu.h:31-51       LAPACK dsyevd('V','U') per block, in place
 *   - static contiguous split over T host threads, the last thread takes the rest, then one-by-one balancing
 *                                        src/duo_solver.cu:344-371, worker loop :598-606
 *   - vector_to_matrices / max(W,0) / V diag(W) / DGEMM(N,T) / matrices_to_vector around it
 *                                        src/kernels/vec_mat_conversion.cu:11-57, dense_scalar.cu:41-47,
 *                                        diagonal_batch.cu:11-23, include/cuadmm/cublas.h:18-35
 * LAPACK / BLAS come from the OpenBLAS that scipy bundles (symbols scipy_dsyevd_, scipy_dgemm_; 32-bit integers), opened
 * with dlopen at run time; BLAS threading is forced to 1 so that the T worker threads are the only parallelism, as in the
 * reference (one dsyevd per thread at a time).
 *
 * engine = 1 replaces dsyevd + DGEMM by the scalar Householder + implicit-QL twin of oracle/eigproj_twin.c (compiled into
 * the same library with -O3 -march=native): the bundled OpenBLAS takes a global buffer lock in every level-2/3 BLAS call,
 * which serialises T threads on small blocks (measured here: 32 x 32 blocks, 1 -> 8 threads: 0.38 s -> 0.23 s; 100 x 100:
 * 0.35 -> 0.075 s).  bench.py times BOTH engines: `cpu_baseline.value` is the faster of the two projection-bound rates (the
 * `eig_engine` field says which), the LAPACK leg -- the reference's own routine -- is always reported beside it (`lapack`), with its
 * single-thread time per block (`lapack_single_thread_us_per_block`) so that a reader sees the lock's serialisation for what it is.
 */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <math.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

typedef void (*dsyevd_fn)(const char*, const char*, const int*, double*, const int*, double*, double*, const int*, int*,
                          const int*, int*);
typedef void (*dgemm_fn)(const char*, const char*, const int*, const int*, const int*, const double*, const double*,
                         const int*, const double*, const int*, const double*, double*, const int*);
typedef void (*setthreads_fn)(int);

static dsyevd_fn p_dsyevd = NULL;
static dgemm_fn p_dgemm = NULL;

static const double kSqrt2 = 0x1.6a09e667f3bccp+0;    /* reference SQRT2    (include/cuadmm/kernels.h:180) */
static const double kSqrt2Inv = 0x1.6a09e667f3bcdp-1; /* reference SQRT2INV (include/cuadmm/kernels.h:181) */

/* returns 0 on success; libpath = the OpenBLAS shared object bundled with scipy */
int cpu_eig_init(const char* libpath) {
  void* h = dlopen(libpath, RTLD_NOW | RTLD_LOCAL);
  if (!h) { fprintf(stderr, "cpu_eig_init: %s\n", dlerror()); return -1; }
  const char* ev[] = {"scipy_dsyevd_", "dsyevd_", NULL};
  const char* gm[] = {"scipy_dgemm_", "dgemm_", NULL};
  const char* th[] = {"scipy_openblas_set_num_threads", "openblas_set_num_threads", "scipy_goto_set_num_threads", NULL};
  for (int i = 0; ev[i] && !p_dsyevd; ++i) p_dsyevd = (dsyevd_fn)dlsym(h, ev[i]);
  for (int i = 0; gm[i] && !p_dgemm; ++i) p_dgemm = (dgemm_fn)dlsym(h, gm[i]);
  setthreads_fn st = NULL;
  for (int i = 0; th[i] && !st; ++i) st = (setthreads_fn)dlsym(h, th[i]);
  if (st) st(1);
  return (p_dsyevd && p_dgemm) ? 0 : -2;
}

int twin_psd_project_block(const double* xin, double* xout, int n);   /* oracle/eigproj_twin.c */

typedef struct {
  const double* in;
  double* out;
  const int* blk;
  const long long* off;
  int first, last;      /* block range of this thread */
  int eig_only;         /* 1: only the eigendecompositions (what the reference runs on the host) */
  int engine;           /* 0: LAPACK dsyevd + DGEMM, 1: the scalar tridiagonal-QL twin */
  int fail;
} job_t;

static void* worker(void* arg) {
  job_t* j = (job_t*)arg;
  if (j->engine == 1) {
    for (int k = j->first; k < j->last; ++k)
      if (twin_psd_project_block(j->in + j->off[k], j->out + j->off[k], j->blk[k])) j->fail++;
    return NULL;
  }
  int nmax = 0;
  for (int k = j->first; k < j->last; ++k) if (j->blk[k] > nmax) nmax = j->blk[k];
  if (nmax == 0) return NULL;
  const int lwork = 1 + 6 * nmax + 2 * nmax * nmax, liwork = 3 + 5 * nmax;
  double* M = (double*)malloc(sizeof(double) * (size_t)nmax * nmax);
  double* T = (double*)malloc(sizeof(double) * (size_t)nmax * nmax);
  double* P = (double*)malloc(sizeof(double) * (size_t)nmax * nmax);
  double* W = (double*)malloc(sizeof(double) * (size_t)nmax);
  double* work = (double*)malloc(sizeof(double) * (size_t)lwork);
  int* iwork = (int*)malloc(sizeof(int) * (size_t)liwork);
  for (int k = j->first; k < j->last; ++k) {
    const int n = j->blk[k];
    const double* x = j->in + j->off[k];
    /* vector_to_matrices: slot order for i = 0..n-1 (column), r = 0..i (row); both triangles, 1/sqrt2 off the diagonal */
    long long e = 0;
    for (int c = 0; c < n; ++c)
      for (int r = 0; r <= c; ++r, ++e) {
        const double v = (r == c) ? x[e] : x[e] * kSqrt2Inv;
        M[(size_t)c * n + r] = v;
        M[(size_t)r * n + c] = v;
      }
    int info = 0;
    p_dsyevd("V", "U", &n, M, &n, W, work, &lwork, iwork, &liwork, &info);   /* eig_cpu.h:38-50 */
    if (info) j->fail++;
    if (j->eig_only) continue;
    /* max(W,0); T = V diag(W+); P = T V^T (column-major, N,T as cublas.h:27-33) */
    for (int c = 0; c < n; ++c) {
      const double w = W[c] > 0.0 ? W[c] : 0.0;
      for (int r = 0; r < n; ++r) T[(size_t)c * n + r] = M[(size_t)c * n + r] * w;
    }
    const double one = 1.0, zero = 0.0;
    p_dgemm("N", "T", &n, &n, &n, &one, T, &n, M, &n, &zero, P, &n);
    /* matrices_to_vector: reads the upper element, sqrt2 off the diagonal */
    double* y = j->out + j->off[k];
    e = 0;
    for (int c = 0; c < n; ++c)
      for (int r = 0; r <= c; ++r, ++e) y[e] = (r == c) ? P[(size_t)c * n + r] : P[(size_t)c * n + r] * kSqrt2;
  }
  free(M); free(T); free(P); free(W); free(work); free(iwork);
  return NULL;
}

/* Projects all blocks (svec layout, blk[nblk] sizes) on `threads` host threads; returns the wall seconds of the parallel
 * region, < 0 on error.  eig_only = 1 times only the dsyevd calls (out untouched). */
double cpu_psd_project(const double* in, double* out, const int* blk, int nblk, int threads, int eig_only, int engine) {
  if (engine == 0 && (!p_dsyevd || !p_dgemm)) return -1.0;
  if (threads < 1) threads = 1;
  if (threads > nblk && nblk > 0) threads = nblk;
  long long* off = (long long*)malloc(sizeof(long long) * ((size_t)nblk + 1));
  off[0] = 0;
  for (int k = 0; k < nblk; ++k) off[k + 1] = off[k] + (long long)blk[k] * (blk[k] + 1) / 2;
  /* duo_solver.cu:344-371: floor(nblk / T) each, the last thread takes the rest, then balance one by one */
  int* per = (int*)calloc((size_t)threads, sizeof(int));
  const int base = nblk / threads;
  for (int t = 0; t < threads - 1; ++t) per[t] = base;
  per[threads - 1] = nblk - (threads - 1) * base;
  if (threads > 2) {
    int i = 0;
    while (i < threads - 1 && per[threads - 1] - per[i] >= 2) { per[i] += 1; per[threads - 1] -= 1; ++i; }
  }
  job_t* jobs = (job_t*)calloc((size_t)threads, sizeof(job_t));
  pthread_t* th = (pthread_t*)calloc((size_t)threads, sizeof(pthread_t));
  int sum = 0;
  for (int t = 0; t < threads; ++t) {
    jobs[t].in = in; jobs[t].out = out; jobs[t].blk = blk; jobs[t].off = off;
    jobs[t].first = sum; sum += per[t]; jobs[t].last = sum; jobs[t].eig_only = eig_only; jobs[t].engine = engine; jobs[t].fail = 0;
  }
  struct timespec t0, t1;
  clock_gettime(CLOCK_MONOTONIC, &t0);
  for (int t = 1; t < threads; ++t) pthread_create(&th[t], NULL, worker, &jobs[t]);
  worker(&jobs[0]);
  for (int t = 1; t < threads; ++t) pthread_join(th[t], NULL);
  clock_gettime(CLOCK_MONOTONIC, &t1);
  int fail = 0;
  for (int t = 0; t < threads; ++t) fail += jobs[t].fail;
  free(off); free(per); free(jobs); free(th);
  if (fail) return -2.0;
  return (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
}
