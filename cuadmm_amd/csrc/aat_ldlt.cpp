// Host factor + permuted solve of (A A^T + eps I): the engine's replacement for the reference's
// CholeskySolverCPU (include/cuadmm/cholesky_cpu.h:62-155), which wraps CHOLMOD's simplicial
// LDL^T with `cholmod_analyze` ordering and `cholmod_solve2(CHOLMOD_LDLt)`.
//
// SuiteSparse is not available to this build, so the three pieces are implemented here:
//   1. B = A A^T + eps I by row-wise sparse accumulation,
//   2. a fill-reducing ordering: quotient-graph approximate minimum degree with element
//      absorption and aggressive absorption,
//   3. an up-looking sparse LDL^T (elimination tree + row-pattern reach) and the two
//      triangular solves.
// Contract kept from the reference: the solve works on the PERMUTED system and applies no
// permutation itself (cholesky_cpu.h:146-155; solver.cu:487,500 permute around it).
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <numeric>
#include <thread>

#if defined(__linux__)
#include <sched.h>
#endif

#include "common.h"

struct cuadmm_aat {
  int m = 0;
  std::vector<int> perm, iperm;
  std::vector<int64_t> Lp;  // column pointers of unit-lower L (strict part), m+1
  std::vector<int> Li;
  std::vector<double> Lx;
  std::vector<double> D;
  // split factorisation (cuadmm_aat_create_split): the last tail_k columns are NOT factored on the host; schur_*
  // hold the lower triangle (with diagonal) of the Schur complement B22 - L21 D1 L21^T by rows (CSR, tail-local
  // column indices) for the GPU.  Sparse on purpose: a dense 17 152^2 host matrix costs seconds of page faults.
  // Independent subtrees of the elimination tree, grouped into chunks of columns (ascending within a chunk): the
  // sweeps of different chunks touch disjoint entries, so cuadmm_aat_solve_permuted runs them on a few host threads
  // with bit-identical results (used for large block-diagonal systems, e.g. weak-scaled runs with m = 400 000).
  std::vector<std::vector<int>> chunks;
  // Split factor: the trees of the LEADING columns' elimination forest grouped into kLeadChunks chunks (columns ascending within a
  // chunk).  The forward sweep of a chunk updates its own leading entries and a tail accumulator of its own (length tail_k); the
  // accumulators are added to the tail in chunk order (deterministic, independent of the thread count); the backward sweep of a
  // chunk reads the solved tail and its own entries.  For forests of many trees whose sweeps are milliseconds on one core (PushBox
  // N = 30: 4 953 trees, deepest 1 135 levels -- too deep for the device-side sweeps --, 1.6 ms per sweep serial).
  std::vector<std::vector<int>> lead_chunks;
  std::vector<int64_t> lead_mid;             // split factor: first entry of leading column j in a TAIL row (rows ascend inside a column)
  mutable std::vector<double> lead_acc;      // kLeadChunks x tail_k: scratch of cuadmm_aat_solve_leading_forward.  SINGLE CALLER: two threads
                                             // sweeping the same factor at once would share it (the engine has one solve in flight per handle;
                                             // include/cuadmm_amd.h says so at the entry point)
  std::vector<int> nzcols;   // columns j < m - tail_k with at least one sub-diagonal entry, ascending (block-diagonal A A^T: few)
  int tail_k = 0;
  int plan_tops = 0;         // > 0: plan_tail chose this tail FOR the device-side solve with dense tree tops cut at this height (lead_solve.h)
  std::vector<int64_t> schur_ptr;
  std::vector<int> schur_col;
  std::vector<double> schur_val;
  double analyze_s = 0, factor_s = 0;
  // elimination forest as lists of columns per tree (cuadmm_aat_forest): tree t owns forest_cols[forest_ptr[t] .. forest_ptr[t+1]),
  // ascending; the sweeps of a solve never leave a tree
  std::vector<int> forest_ptr, forest_cols;
  int forest_max = 0;
};

namespace {

using cuadmm::set_error;

void min_degree_order_core(int n, const std::vector<int64_t>& Bp, const std::vector<int>& Bi, std::vector<int>& perm);

// ---------------------------------------------------------------------------------------
// Approximate minimum degree on the pattern of a symmetric matrix (diagonal ignored).
// Quotient graph: variables keep a list of adjacent variables (edges not yet covered by an
// element) and a list of adjacent elements; an element keeps its variable list.
// ---------------------------------------------------------------------------------------
void min_degree_order(int n_all, const std::vector<int64_t>& Bp, const std::vector<int>& Bi, std::vector<int>& perm) {
  perm.resize(n_all);
  // Dense rows first (the rule of AMD, Amestoy / Davis / Duff): a row with more than max(16, 10 sqrt(n)) entries takes no part in
  // the elimination graph and is ordered last, by degree.  PushT_N=30: 8 560 of the 53 290 rows of A A^T hold 74 of its 77 M
  // nonzeros (8 600 entries each); they end up in the dense tail whatever the order, and carrying them through the quotient
  // graph cost 26 of the 36 s of the analysis.  Inputs without such rows (every other fixture) keep their ordering bit for bit.
  const long long dense_thr = std::max<long long>(16, (long long)(10.0 * std::sqrt((double)n_all)));
  std::vector<char> dense(n_all, 0);
  int n_dense = 0;
  int64_t nnz_dense = 0;
  for (int i = 0; i < n_all; ++i)
    if (Bp[i + 1] - Bp[i] - 1 > dense_thr) { dense[i] = 1; ++n_dense; nnz_dense += Bp[i + 1] - Bp[i]; }
  // ... applied where carrying them is what the ordering costs: the dense rows hold most of a LARGE pattern (>= 16 M nonzeros).
  // Smaller inputs keep the plain ordering (PushT_N=10: its GPU tail reproduces the oracle's pobj to 7e-9 with it, 2.8e-8 with
  // the dense rows last -- same solve, another composition of the numerically singular tail).
  if (n_dense > 0 && (Bp[n_all] < 16000000 || 2 * nnz_dense < Bp[n_all])) n_dense = 0;
  if (n_dense > 0) {
    // the sparse part is ordered on its own graph (dense neighbours dropped); recursion depth one
    std::vector<int> map_new(n_all, -1), map_old;
    for (int i = 0; i < n_all; ++i) if (!dense[i]) { map_new[i] = (int)map_old.size(); map_old.push_back(i); }
    const int ns = (int)map_old.size();
    std::vector<int64_t> Sp((size_t)ns + 1, 0);
    std::vector<int> Si;
    for (int q = 0; q < ns; ++q) {
      const int i = map_old[q];
      for (int64_t p = Bp[i]; p < Bp[i + 1]; ++p) if (!dense[Bi[p]]) Si.push_back(map_new[Bi[p]]);
      Sp[(size_t)q + 1] = (int64_t)Si.size();
    }
    std::vector<int> sperm;
    min_degree_order_core(ns, Sp, Si, sperm);                 // one level: what is dense relative to the sparse part stays in it
    for (int q = 0; q < ns; ++q) perm[q] = map_old[sperm[q]];
    std::vector<int> dl;
    for (int i = 0; i < n_all; ++i) if (dense[i]) dl.push_back(i);
    std::stable_sort(dl.begin(), dl.end(), [&](int a, int b) { return Bp[a + 1] - Bp[a] < Bp[b + 1] - Bp[b]; });
    for (size_t q = 0; q < dl.size(); ++q) perm[(size_t)ns + q] = dl[q];
    return;
  }
  min_degree_order_core(n_all, Bp, Bi, perm);
}

void min_degree_order_core(int n, const std::vector<int64_t>& Bp, const std::vector<int>& Bi, std::vector<int>& perm) {
  perm.resize(n);
  std::vector<std::vector<int>> adj(n), elems(n), evars(n);
  for (int i = 0; i < n; ++i) {
    adj[i].reserve((size_t)(Bp[i + 1] - Bp[i]));
    for (int64_t p = Bp[i]; p < Bp[i + 1]; ++p)
      if (Bi[p] != i) adj[i].push_back(Bi[p]);
  }
  // state: 0 live variable, 1 live element, 2 dead (absorbed element)
  std::vector<char> state(n, 0);
  std::vector<int> deg(n), head(n + 1, -1), nxt(n, -1), prv(n, -1);
  auto bucket_insert = [&](int v, int d) {
    deg[v] = d; prv[v] = -1; nxt[v] = head[d];
    if (head[d] >= 0) prv[head[d]] = v;
    head[d] = v;
  };
  auto bucket_remove = [&](int v) {
    int d = deg[v];
    if (prv[v] >= 0) nxt[prv[v]] = nxt[v]; else head[d] = nxt[v];
    if (nxt[v] >= 0) prv[nxt[v]] = prv[v];
  };
  for (int i = 0; i < n; ++i) bucket_insert(i, (int)adj[i].size());
  std::vector<int> mark(n, -1), wstamp(n, -1), wcnt(n, 0), Lp;
  int mindeg = 0;
  for (int k = 0; k < n; ++k) {
    while (head[mindeg] < 0) ++mindeg;
    // What is left is (nearly) a clique: the smallest external degree reaches 70 % of the remaining variables.  Any order of
    // them fills the trailing block completely -- it becomes the dense tail the GPU factors -- while the quotient-graph updates of
    // this phase are the expensive ones (every pivot walks thousands of long lists: PushT_N=30, A A^T with 77 M nonzeros on
    // m = 53 290, spent 36 of its 45 s of analysis here).  The rest is ordered by current degree.
    // `mindeg` is the APPROXIMATE degree (an upper bound: |adj| + |L_p| - 1 + the external parts of the other elements, which may
    // overlap), so the test is confirmed with the exact external degree of the candidate -- the distinct live variables it reaches --
    // before the ordering is cut short; an approximate degree that overshoots must not end it while the rest is far from a clique.
    if (n - k > 2048 && (long long)mindeg * 10 >= 7LL * (n - k - 1)) {
      const int c = head[mindeg];
      const int stamp = -2 - k;                     // a stamp of its own: `mark` holds step indices >= 0 (and -1)
      long long exact = 0;
      mark[c] = stamp;
      for (int v : adj[c])
        if (state[v] == 0 && mark[v] != stamp) { mark[v] = stamp; ++exact; }
      for (int e : elems[c]) {
        if (state[e] != 1) continue;
        for (int v : evars[e])
          if (state[v] == 0 && mark[v] != stamp) { mark[v] = stamp; ++exact; }
      }
      if (exact * 10 >= 7LL * (n - k - 1)) {
        for (int d = mindeg; d <= n && k < n; ++d)
          for (int v = head[d]; v >= 0; v = nxt[v]) perm[k++] = v;
        break;
      }
    }
    int p = head[mindeg];
    bucket_remove(p);
    perm[k] = p;
    // variables of the new element p
    Lp.clear();
    mark[p] = k;
    for (int v : adj[p])
      if (state[v] == 0 && mark[v] != k) { mark[v] = k; Lp.push_back(v); }
    for (int e : elems[p]) {
      if (state[e] != 1) continue;
      for (int v : evars[e])
        if (mark[v] != k) { mark[v] = k; Lp.push_back(v); }
      state[e] = 2;
      std::vector<int>().swap(evars[e]);
    }
    state[p] = 1;
    std::vector<int>().swap(adj[p]);
    std::vector<int>().swap(elems[p]);
    // |L_e ∩ L_p| for every live element touching L_p
    for (int i : Lp)
      for (int e : elems[i]) {
        if (state[e] != 1) continue;
        if (wstamp[e] != k) { wstamp[e] = k; wcnt[e] = 0; }
        wcnt[e]++;
      }
    const int lp_size = (int)Lp.size();
    for (int i : Lp) {
      bucket_remove(i);
      // edges to members of L_p are now represented by element p
      auto& a = adj[i];
      size_t w = 0;
      for (size_t t = 0; t < a.size(); ++t) {
        int v = a[t];
        if (state[v] == 0 && mark[v] != k) a[w++] = v;
      }
      a.resize(w);
      long long d = (long long)w + (lp_size - 1);
      auto& el = elems[i];
      w = 0;
      for (size_t t = 0; t < el.size(); ++t) {
        int e = el[t];
        if (state[e] != 1) continue;
        int ext = (int)evars[e].size() - wcnt[e];
        if (ext <= 0) {  // e is a subset of L_p: aggressive absorption
          state[e] = 2;
          std::vector<int>().swap(evars[e]);
          continue;
        }
        d += ext;
        el[w++] = e;
      }
      el.resize(w);
      el.push_back(p);
      long long bound = (long long)deg[i] + (lp_size - 1);
      if (d > bound) d = bound;
      if (d > n - k - 1) d = n - k - 1;
      if (d < 0) d = 0;
      bucket_insert(i, (int)d);
      if ((int)d < mindeg) mindeg = (int)d;
    }
    evars[p] = Lp;
  }
}

double now_s() {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// Minimal fork-join pool.  run(f) executes f(0..T-1), f(0) on the caller.
// Hand-off is spin-then-sleep: a worker polls the generation counter for kSpinUs after its last job before it blocks on the
// condition variable, and the caller polls the completion counter.  Inside an ADMM solve the host phases come every few
// hundred microseconds, so the workers stay hot and a fork-join costs ~1 us instead of the ~50 us of a condition-variable
// wake-up (which made threads a loss below m = 200 000 in round 1); an idle pool sleeps.
// The pool is a process-wide singleton and run() is a single-occupancy fork-join (job / pending / gen belong to the one
// call in flight), so concurrent callers (two engines driven from two host threads, tests/test_gpu_sharded.py) are
// serialised by `run_mu`: the caller that finds the pool busy runs its chunks inline instead of waiting -- the chunks'
// results do not depend on which thread runs them.
class HostPool {
 public:
  explicit HostPool(int threads) : T(std::max(1, threads)) {
    for (int i = 1; i < T; ++i) workers.emplace_back([this, i] { loop(i); });
  }
  ~HostPool() {
    { std::lock_guard<std::mutex> lk(mu); stop.store(true); }
    cv_start.notify_all();
    for (auto& t : workers) t.join();
  }
  int size() const { return T; }
  void run(const std::function<void(int)>& f) {
    if (T == 1) { f(0); return; }
    std::unique_lock<std::mutex> occupancy(run_mu, std::try_to_lock);
    if (!occupancy.owns_lock()) {
      for (int i = 0; i < T; ++i) f(i);
      return;
    }
    job = &f;
    pending.store(T - 1, std::memory_order_relaxed);
    // store-buffer pattern with the worker's (sleepers++ ; read gen): both sides sequentially consistent, or on a weakly
    // ordered host each could miss the other (worker asleep, no notify, caller spinning on `pending` for ever)
    gen.fetch_add(1, std::memory_order_seq_cst);
    if (sleepers.load(std::memory_order_seq_cst) > 0) {
      std::lock_guard<std::mutex> lk(mu);      // pairs with the predicate check of a worker about to block
      cv_start.notify_all();
    }
    f(0);
    for (int spins = 0; pending.load(std::memory_order_acquire) != 0; ++spins) {
      if (spins < 20000) cpu_relax();
      else std::this_thread::yield();
    }
    job = nullptr;
  }

 private:
  static void cpu_relax() {
#if defined(__x86_64__) || defined(__i386__)
    __builtin_ia32_pause();
#endif
  }
  static constexpr double kSpinUs = 2000.0;
  void loop(int id) {
    unsigned seen = 0;
    for (;;) {
      // poll, then block
      const auto t0 = std::chrono::steady_clock::now();
      bool got = false;
      for (int it = 0;; ++it) {
        if (gen.load(std::memory_order_acquire) != seen) { got = true; break; }
        if (stop.load(std::memory_order_relaxed)) return;
        cpu_relax();
        if ((it & 255) == 255 &&
            std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() > kSpinUs) break;
      }
      if (!got) {
        std::unique_lock<std::mutex> lk(mu);
        sleepers.fetch_add(1, std::memory_order_seq_cst);
        cv_start.wait(lk, [&] { return stop.load() || gen.load(std::memory_order_seq_cst) != seen; });
        sleepers.fetch_sub(1, std::memory_order_acq_rel);
        if (stop.load()) return;
      }
      seen = gen.load(std::memory_order_acquire);
      const std::function<void(int)>* j = job;
      (*j)(id);
      pending.fetch_sub(1, std::memory_order_release);
    }
  }
  const int T;
  std::vector<std::thread> workers;
  std::mutex mu, run_mu;
  std::condition_variable cv_start;
  const std::function<void(int)>* job = nullptr;
  std::atomic<unsigned> gen{0};
  std::atomic<int> pending{0}, sleepers{0};
  std::atomic<bool> stop{false};
};

// CPUs this process may really use: the affinity mask capped by the cgroup CPU quota (the GPU boxes show 256 hardware threads
// under a 16-CPU quota: eight ranks with eight spinning threads each would be 64 busy threads on 16 CPUs).
static int usable_cpus() {
  int n = (int)std::thread::hardware_concurrency();
#if defined(__linux__)
  cpu_set_t set;
  CPU_ZERO(&set);
  if (sched_getaffinity(0, sizeof set, &set) == 0) { const int c = CPU_COUNT(&set); if (c > 0) n = c; }
  if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
    char quota[64] = {0};
    long long period = 0;
    if (fscanf(f, "%63s %lld", quota, &period) == 2 && quota[0] != 'm' && period > 0) {
      const long long q = atoll(quota);
      if (q > 0) n = std::min<long long>(n, std::max<long long>(1, (q + period / 2) / period));
    }
    fclose(f);
  }
#endif
  return std::max(1, n);
}

// Threads of the host pool: CUADMM_HOST_THREADS if set (1 = serial), else min(16, usable CPUs / ranks on this node) (16 since the
// largest inputs of round 4: PlanarHand_N=10's L11 sweeps 24.8 -> 10.0 ms per iteration from 8 to 16 threads on the 16-CPU box) -- the
// launcher's LOCAL_WORLD_SIZE (torch.distributed.run) or cuadmm_host_pool_hint() say how many ranks share the node.  Created on
// first use, lives until process exit.
static std::atomic<int> g_pool_ranks_hint{0};
HostPool& host_pool() {
  static HostPool pool([] {
    const char* e = getenv("CUADMM_HOST_THREADS");
    if (e) return std::max(1, atoi(e));
    int ranks = g_pool_ranks_hint.load();
    if (ranks <= 0) { const char* l = getenv("LOCAL_WORLD_SIZE"); ranks = l ? atoi(l) : 1; }
    ranks = std::max(1, ranks);
    return std::max(1, std::min(16, usable_cpus() / ranks));
  }());
  return pool;
}

// Cost model of the dense-tail split: host 1.2 ns per nonzero of the leading columns (both sweeps, measured),
// GPU 60 us + k^2 * 8 B at 4 TB/s.  Returns 0 when the split does not pay for its PCIe round trip.
// With the elimination tree (parent) the choice is refined for the DEVICE-side solve the engine runs when the leading forest is
// shallow (lead_solve.hip): both sweeps cost ~0.65 us per level of the deepest leading tree (plus a side stream while big trees
// exist), the tail one pass over k^2 / 2 doubles (two GEMVs beyond 20 480 columns) -- a somewhat larger tail that halves the depth
// wins (pendulum N = 80: k 9 728 -> 12 032, depth 61 -> 28, y-solve 0.26 -> 0.20 ms).  The height of the forest restricted to the
// first n1 columns is a prefix maximum of the node heights (a parent always has the larger index): one pass for every k.
// Round 5, DENSE TREE TOPS (lead_solve.h): with the nodes of height >= 32 of every leading tree solved through explicit inverses of their
// diagonal blocks, the depth of the forest stops deciding the tail.  What a larger tail bought -- swallowing the long chains -- the tops do at
// a fraction of the bytes: per solve the two passes over the packed inverses (measured ~4 TB/s), two SpMVs over what leaves the shallow rest,
// 2 x 32 levels of sweeps, and a tail of HALF the size or less.  Measured per sGS iteration (profiles/r05_tops_scan.log): PushBox N = 30
// k 18 688 -> 4 096 ... 10 240: 1.81 -> 0.96 ... 1.01 ms; N = 50 k 30 720 -> 8 192: 2.93 -> 1.16; PushT_N=30 27 136 -> 16 384: 2.44 -> 1.09;
// PlanarHand_N=1 17 152 -> 8 192: 0.90 -> 0.75; PlanarHand_N=10 keeps its 32 768 columns (smaller tails grow 2 ... 5 GB of tops) but leaves
// the host: 14.5 -> 4.7.  `tops_out` receives the height of the cut when the returned tail is meant for that solve (0 otherwise);
// `allow_tops` = false is the planner of round 4.
thread_local bool g_plan_allow_tops = true;
int plan_tail(const int64_t* Lp, int m, int max_k, const int* parent = nullptr, int* tops_out = nullptr) {
  if (tops_out) *tops_out = 0;
  const double host_ns = 1.2, total = host_ns * (double)Lp[m];
  // Small systems: a host solve is its nonzeros plus two PCIe hops and a stream synchronisation (~35 us); the WHOLE factor as a dense
  // tail on the device is one pass over 8 m^2 bytes of explicit inverse behind a right-hand side that never leaves HBM (lead_solve.hip
  // with no leading columns, ~40 us of launches).  Measured per sGS iteration, host -> device: taha1a (m = 3 002, 159 k nonzeros) 1.12 ->
  // 0.86 ms, swissroll (3 380, 101 k) 3.59 -> 3.42, biggs (1 819, 42 k) 0.50 -> 0.48; rose13 (2 379, DIAGONAL A A^T) must stay where it
  // is (forest solve on the device: 0.36 ms against 0.43 through a dense tail) -- and does by this model.
  if (parent && m >= 512 && m <= 4096 && max_k >= m) {
    const double dev_ns = 40e3 + (double)m * m * 8.0 / 4000.0;
    if (total + 35e3 > 1.5 * dev_ns) return m;
  }
  double best = total;
  int best_k = 0;
  for (int k = 256; k <= std::min(m, max_k); k += 256) {
    const double cost = host_ns * (double)Lp[m - k] + 60e3 + (double)k * k * 8.0 / 4000.0;
    if (cost < best) { best = cost; best_k = k; }
  }
  if (best_k == 0 || best > 0.7 * total || total < 300e3) {
    // WIDE FOREST (round 5): no dense tail pays for itself -- the factor is sparse to the top -- but the host still spends a millisecond per
    // solve on it (bqp-r1-40-1: m = 269 001, 0.8 M nonzeros, 1.0 of 4.2 ms per iteration with its two PCIe hops).  With a SMALL tail in front of the
    // device-side sweeps the whole solve stays in HBM: 103 016 trees there, 102 000 of them of one or two nodes (a thread each:
    // lead_solve.h, micro trees), the rest 105 levels deep.  Taken when every tree fits the sweeps (6 144 nodes, 256 levels), the model says
    // half the host's time or less, and the forest is not the block-diagonal kind the engine solves with one thread per tree anyway.
    if (parent && g_plan_allow_tops && m >= 65536 && total >= 500e3) {
      for (const int k : {1024, 2048, 4096}) {
        if (k > std::min(m - 1, max_k)) break;
        const int n1 = m - k;
        std::vector<int> sz((size_t)n1, 1), ht((size_t)n1, 1);
        int big = 0, deep = 0, big0 = 0;
        long long in_trees = 0;
        for (int j = 0; j < n1; ++j) {
          const int p = parent[j];
          if (p >= 0 && p < n1) { sz[p] += sz[j]; ht[p] = std::max(ht[p], ht[j] + 1); }
          else { big = std::max(big, sz[j]); deep = std::max(deep, ht[j]); if (sz[j] > 2) in_trees += sz[j]; }
        }
        {
          std::vector<int> s0((size_t)m, 1);
          for (int j = 0; j < m; ++j) { const int p = parent[j]; if (p >= 0) s0[p] += s0[j]; else big0 = std::max(big0, s0[j]); }
        }
        if (big0 <= 64 || big > 6144 || deep > 256) continue;
        const double dev_us = 2.0 * (14.0 + 0.25e-3 * (double)in_trees) + 16.0 + 45.0 + 24.0 * (double)Lp[n1] / 3.0e6 + (double)k * k * 4.0 / 5.1e6 + 28.0;
        // with the tree tops: the one tree that does not fit a workgroup's LDS with its stream resident (1 681 nodes, 105 levels on the
        // streaming kernel: 0.8 ms per solve, slower than the host) is cut like every other -- measured 1.33 -> 0.19 ms of y-solve per
        // iteration, 4.45 -> 3.36 ms per iteration
        if (dev_us < 0.5 * (total * 1e-3 + 150.0)) { if (tops_out) *tops_out = 32; return k; }
      }
    }
    return 0;
  }
  if (!parent) return best_k;
  std::vector<int> h((size_t)m, 1), hmax((size_t)m + 1, 0);      // hmax[n1] = height of the forest over columns < n1
  for (int j = 0; j < m; ++j) {
    hmax[(size_t)j + 1] = std::max(hmax[j], h[j]);
    const int p = parent[j];
    if (p >= 0 && h[p] < h[j] + 1) h[p] = h[j] + 1;
  }
  // the device-side solve with dense tree tops at tail size k, the forest cut at height lvl: model in us per solve, < 0 when it cannot be built.
  // Round 6: recalibrated on the kernel times of profiles/r05_c1_kernel_stats.csv and this round's bench lines -- the first model charged the
  // sweeps 2 (20 + 0.9 lvl) = 98 us at height 32 where PlanarHand_N=1 measures 2 x 25, and 70 us of short kernels where there are 55; it kept
  // pendulum N = 80 on its plain plan (176 against 166 us by the models; 123 against 171 measured at 6 144 columns and height 16).  The cut is now
  // chosen from {16, 32} by the same model (round 5 fixed 32: right for the deep forests of PushBox / PlanarHand_N=10, 1 - 4 % behind 16 on the
  // shallow ones) -- measured picks: profiles/r06_plan_picks.log.
  auto tops_us = [&](int k, int lvl) {
    const int n1 = m - k;
    if (n1 <= 0) return -1.0;
    // roots of the leading forest; sizes of the tops (nodes of height > lvl: h counts from 1) and of the rest's trees
    std::vector<int> root((size_t)n1), tcnt((size_t)n1, 0), bcnt((size_t)n1, 0), broot((size_t)n1);
    int nT = 0;
    for (int j = n1 - 1; j >= 0; --j) {
      const int p = parent[j];
      const bool top = h[j] > lvl;
      root[j] = (p < 0 || p >= n1) ? j : root[p];
      if (top) { tcnt[root[j]]++; ++nT; }
      else { broot[j] = (p < 0 || p >= n1 || h[p] > lvl) ? j : broot[p]; bcnt[broot[j]]++; }
    }
    if (nT == 0) return -1.0;
    double tri = 0.0;
    for (int j = 0; j < n1; ++j) {
      if (bcnt[j] > 6144) return -1.0;                              // lead_solve.hip: one tree per workgroup
      tri += 0.5 * (double)tcnt[j] * ((double)tcnt[j] + 1.0);
    }
    if (tri * 16.0 > (double)cuadmm::kLeadTopsMaxBytes) return -1.0;                            // both triangles: memory, and the n^3-ish build on the host
    // the tail's one pass with 16-byte loads (round 6): ~5.3 TB/s + its reduction up to 18 432 columns, the row-sharing kernel beyond
    const double tail = k <= 18432 ? (double)k * k * 4.0 / 5.3e6 + 12.0 : (double)k * k * 4.0 / (k <= 24576 ? 4.2e6 : 5.0e6) + 40.0;
    // both sweeps over the shallow rest (a wavefront per tree, lvl levels; beyond ~100 000 nodes the launch runs in rounds), the short kernels
    // of the middle stage, both dense passes over the packed inverses, the two SpMVs over what leaves the rest
    const double nodes_rest = (double)(n1 - nT);
    const double sweeps = 2.0 * (12.0 + 0.45 * lvl + 0.2e-3 * std::max(0.0, nodes_rest - 100e3));
    return sweeps + 55.0 + tri * 16.0 / 4.4e6 + 24.0 * (double)Lp[n1] / 3.0e6 + tail;
  };
  // the best tail for that solve: not below half the host optimum (the host factors what the tail does not)
  auto plan_tops = [&](int k_lo, double other_us) {
    if (!g_plan_allow_tops || !tops_out) return 0;
    const bool plan_debug = getenv("CUADMM_PLAN_DEBUG") != nullptr;      // developer aid: the model's figures on stderr
    int kb = 0, lb = 0;
    double cb = 1e300;
    std::vector<std::pair<std::pair<int, int>, double>> cand;
    // (multiples of 1 024: the tail's one-pass kernel holds whole columns per thread of its 1 024-thread workgroups -- 8 448 columns are
    // padded to 9 216)
    for (const int lvl : {32, 16})
      for (int k = std::max(1024, k_lo / 2 / 1024 * 1024); k <= std::min(m - 1, max_k); k += 1024) {
        if (hmax[(size_t)(m - k)] <= lvl + 1) break;                // nothing left to cut
        const double c = tops_us(k, lvl);
        if (plan_debug) fprintf(stderr, "[plan debug] tops: k %d cut %d height %d model %.0f us (other plan %.0f)\n", k, lvl, hmax[(size_t)(m - k)], c, other_us);
        if (c < 0.0) continue;
        cand.push_back({{k, lvl}, c});
        if (c < cb) { cb = c; kb = k; lb = lvl; }
      }
    if (kb == 0 || !(cb < 0.9 * other_us)) return 0;
    // the model is good to ~10 %: of the tails within 5 % of the best AT THE SAME CUT the largest (less for the host to factor, smaller tops)
    for (const auto& kc : cand) if (kc.first.second == lb && kc.second <= 1.05 * cb) kb = std::max(kb, kc.first.first);
    *tops_out = lb;
    return kb;
  };
  auto dev_us = [&](int k) {
    const double ht = (double)hmax[(size_t)(m - k)];
    const double tail = k <= 20480 ? (double)k * k * 4.0 / 5.3e6 + 12.0 : (double)k * k * 4.0 / 4.2e6 + 40.0;     // (round 6: 16-byte loads; the row-sharing kernel beyond)
    return 2.0 * (8.0 + 0.65 * ht) + (ht > 40.0 ? 25.0 : 0.0) + 20.0 + tail;                                     // + the right-hand side, L21^T x2: two short kernels
  };
  // Deep forest at the host optimum: the sweeps stay on the host (two PCIe hops per solve) -- UNLESS a larger tail swallows the long
  // chains.  PushBox N = 30 (m = 154 256): height 1 135 up to k = 17 408, 632 at 17 920, 120 at 18 432, 67 at 19 456; with the device-side
  // sweeps at k = 18 432 the sGS iteration goes 3.66 -> 1.53 ms (measured per solve: one-pass tail 266 us at 5.1 TB/s, the two sweeps
  // 108 us each over 120 levels, L21 products 2 x 28 us; beyond 18 432 columns the tail's right-hand side no longer fits the LDS and
  // four workgroups share a row).  Guards: the larger tail must pay for its k^3 build (measured ~3e13 flop/s) within
  // 3 000 solves, and every leading tree must fit one workgroup's LDS (lead_solve.hip: 6 144 nodes).
  if (hmax[(size_t)(m - best_k)] > 256) {
    const int k_hi = std::min(m, max_k);
    const double host_us = best * 1e-3 + 150.0;
    auto deep_us = [&](int k) {
      const double ht = (double)hmax[(size_t)(m - k)];
      // one workgroup per row up to 18 432 columns (5.1 TB/s), four per row beyond (tail_solve.hip: measured 4.2 / 3.8 TB/s)
      const double tail = k <= 18432 ? (double)k * k * 4.0 / 5.1e6 + 28.0 : (double)k * k * 4.0 / (k <= 24576 ? 4.2e6 : 3.8e6) + 40.0;
      return 2.0 * (35.0 + 0.9 * ht) + 60.0 + tail;
    };
    auto max_tree = [&](int k) {                                   // nodes of the largest leading tree (parents follow their children)
      const int n1 = m - k;
      std::vector<int> sz((size_t)n1, 1);
      int mx = 0;
      for (int j = 0; j < n1; ++j) {
        const int p = parent[j];
        if (p >= 0 && p < n1) sz[p] += sz[j]; else mx = std::max(mx, sz[j]);
      }
      return mx;
    };
    int kd = 0;
    double cd = 1e300;
    for (int k = best_k + 256; k <= k_hi; k += 256) {
      if (hmax[(size_t)(m - k)] > 256) continue;
      const double c = deep_us(k);
      if (c < cd) { cd = c; kd = k; }
    }
    int k_deep = 0;
    if (!(kd == 0 || cd > 0.6 * host_us))
      for (int k = kd; k <= k_hi && deep_us(k) <= 1.15 * cd; k += 256) {
        if (hmax[(size_t)(m - k)] > 256 || max_tree(k) > 6144) continue;
        const double extra_s = ((double)k * k * k - (double)best_k * best_k * best_k) / 3e13;
        if ((host_us - deep_us(k)) * 3000e-6 < extra_s) break;
        k_deep = k;
        break;
      }
    // against the larger tail where one was found, else against the solve the deep forest leaves (device sweeps over every level where
    // the trees fit -- PushT_N=30 --, the host otherwise)
    const double other = k_deep ? deep_us(k_deep) : std::min(host_us, deep_us(best_k));
    if (const int kt = plan_tops(best_k, other)) return kt;
    return k_deep ? k_deep : best_k;
  }
  int k_dev = best_k;
  double c_dev = dev_us(best_k);
  // only LARGER tails are considered: towards smaller ones the leading trees grow long rows, which the per-level figure does not
  // see (PlanarHand_N=1 at k = 12 544 instead of 17 152: trees of 25 000 entries, sweeps three times slower)
  for (int k = best_k + 256; k <= std::min(std::min(m, max_k), 2 * best_k); k += 256) {
    const double c = dev_us(k);
    if (c < 0.93 * c_dev) { c_dev = c; k_dev = k; }               // a clear gain only: the model is good to ~10 %
  }
  if (const int kt = plan_tops(best_k, c_dev)) return kt;
  return k_dev;
}

int aat_create_impl(int m, int L, const int* Acp, const int* Ari, const double* Ax, double eps, int split_max_k, cuadmm_aat** out);

}  // namespace

extern "C" {

int cuadmm_aat_create(int m, int L, const int* Acp, const int* Ari, const double* Ax, double eps, cuadmm_aat** out) {
  return aat_create_impl(m, L, Acp, Ari, Ax, eps, 0, out);
}

// Same, but when the cost model finds a dense tail (k <= max_k) the last k columns are left unfactored and the dense
// Schur complement is kept for the GPU (cuadmm_aat_tail_k / _tail_schur); cuadmm_aat_solve_permuted is then unavailable.
int cuadmm_aat_create_split(int m, int L, const int* Acp, const int* Ari, const double* Ax, double eps, int max_k, cuadmm_aat** out) {
  return aat_create_impl(m, L, Acp, Ari, Ax, eps, max_k, out);   // max_k < 0 forces the tail size -max_k (tests, CUADMM_TAIL_K)
}

}  // extern "C"

namespace {

int aat_create_impl(int m, int L, const int* Acp, const int* Ari, const double* Ax, double eps, int split_max_k, cuadmm_aat** out) {
  if (!out || m < 0 || L < 0 || !Acp) { set_error("aat_create: bad arguments"); return CUADMM_ERR_INVALID; }
  *out = nullptr;
  double t0 = now_s();
  const int64_t nnz = Acp[L];
  // rows of A (CSR) from its columns (CSC)
  std::vector<int64_t> Rp((size_t)m + 1, 0);
  for (int64_t p = 0; p < nnz; ++p) {
    if (Ari[p] < 0 || Ari[p] >= m) { set_error("aat_create: row index %d out of range", Ari[p]); return CUADMM_ERR_INVALID; }
    Rp[(size_t)Ari[p] + 1]++;
  }
  for (int i = 0; i < m; ++i) Rp[i + 1] += Rp[i];
  std::vector<int> Rc((size_t)nnz);
  std::vector<double> Rx((size_t)nnz);
  {
    std::vector<int64_t> pos(Rp.begin(), Rp.end() - 1);
    for (int k = 0; k < L; ++k)
      for (int64_t p = Acp[k]; p < Acp[k + 1]; ++p) {
        int64_t q = pos[Ari[p]]++;
        Rc[q] = k; Rx[q] = Ax[p];
      }
  }
  // B = A A^T + eps I, full symmetric pattern, one row at a time
  std::vector<int64_t> Bp((size_t)m + 1, 0);
  std::vector<int> Bi;
  std::vector<double> Bx;
  {
    std::vector<int> where(m, -1);
    std::vector<double> acc(m, 0.0);
    std::vector<int> touched;
    for (int i = 0; i < m; ++i) {
      touched.clear();
      for (int64_t p = Rp[i]; p < Rp[i + 1]; ++p) {
        int k = Rc[p];
        double a = Rx[p];
        for (int64_t q = Acp[k]; q < Acp[k + 1]; ++q) {
          int j = Ari[q];
          if (where[j] != i) { where[j] = i; acc[j] = 0.0; touched.push_back(j); }
          acc[j] += a * Ax[q];
        }
      }
      if (where[i] != i) { where[i] = i; acc[i] = 0.0; touched.push_back(i); }
      acc[i] += eps;
      std::sort(touched.begin(), touched.end());
      for (int j : touched) { Bi.push_back(j); Bx.push_back(acc[j]); }
      Bp[i + 1] = (int64_t)Bi.size();
    }
  }
  cuadmm_aat* f = new cuadmm_aat();
  f->m = m;
  double t_b = now_s();
  min_degree_order(m, Bp, Bi, f->perm);
  if (getenv("CUADMM_AAT_TIMING")) fprintf(stderr, "[aat] build B %.3fs (nnz %lld), ordering %.3fs\n", t_b - t0, (long long)Bi.size(), now_s() - t_b);
  f->iperm.resize(m);
  for (int i = 0; i < m; ++i) f->iperm[f->perm[i]] = i;

  // C = P B P^T, upper triangle by columns (column k holds rows i <= k)
  std::vector<int64_t> Cp((size_t)m + 1, 0);
  for (int i = 0; i < m; ++i) {
    int pi = f->iperm[i];
    for (int64_t p = Bp[i]; p < Bp[i + 1]; ++p) {
      int pj = f->iperm[Bi[p]];
      if (pi <= pj) Cp[(size_t)pj + 1]++;
    }
  }
  for (int i = 0; i < m; ++i) Cp[i + 1] += Cp[i];
  std::vector<int> Ci((size_t)Cp[m]);
  std::vector<double> Cx((size_t)Cp[m]);
  {
    std::vector<int64_t> pos(Cp.begin(), Cp.end() - 1);
    for (int i = 0; i < m; ++i) {
      int pi = f->iperm[i];
      for (int64_t p = Bp[i]; p < Bp[i + 1]; ++p) {
        int pj = f->iperm[Bi[p]];
        if (pi <= pj) { int64_t q = pos[pj]++; Ci[q] = pi; Cx[q] = Bx[p]; }
      }
    }
  }
  std::vector<int64_t>().swap(Bp); std::vector<int>().swap(Bi); std::vector<double>().swap(Bx);

  // symbolic: elimination tree and column counts of L
  std::vector<int> parent(m, -1), flag(m, -1);
  std::vector<int64_t> Lnz(m, 0);
  for (int k = 0; k < m; ++k) {
    flag[k] = k;
    for (int64_t p = Cp[k]; p < Cp[k + 1]; ++p) {
      int i = Ci[p];
      while (i < k && flag[i] != k) {
        if (parent[i] < 0) parent[i] = k;
        Lnz[i]++;
        flag[i] = k;
        i = parent[i];
      }
    }
  }
  f->Lp.assign((size_t)m + 1, 0);
  for (int k = 0; k < m; ++k) f->Lp[k + 1] = f->Lp[k] + Lnz[k];
  f->analyze_s = now_s() - t0;
  t0 = now_s();
  const int tail_k = split_max_k > 0 ? plan_tail(f->Lp.data(), m, split_max_k, parent.data(), &f->plan_tops) : std::min(m, -split_max_k);
  const int n1 = m - tail_k;                     // rows / columns >= n1 belong to the unfactored tail
  f->tail_k = tail_k;
  try {
    // the tail columns keep their symbolic counts in Lp (cuadmm_aat_factor_nnz reports the whole factor) but get no
    // storage: Li / Lx end at Lp[n1]
    f->Li.resize((size_t)f->Lp[n1]);
    f->Lx.resize((size_t)f->Lp[n1]);
    if (tail_k > 0) {
      f->schur_ptr.assign((size_t)tail_k + 1, 0);
      f->schur_col.reserve((size_t)(f->Lp[m] - f->Lp[n1]) + (size_t)tail_k);
      f->schur_val.reserve((size_t)(f->Lp[m] - f->Lp[n1]) + (size_t)tail_k);
    }
  } catch (const std::bad_alloc&) {
    set_error("aat_create: factor with %lld nonzeros does not fit in host memory", (long long)f->Lp[m]);
    delete f;
    return CUADMM_ERR_FACTOR;
  }
  f->D.assign(m, 0.0);

  // numeric: up-looking LDL^T, row k of L from the reach of column k of C in the etree
  std::vector<double> Y(m, 0.0);
  std::vector<int> pattern(m);
  std::fill(flag.begin(), flag.end(), -1);
  std::fill(Lnz.begin(), Lnz.end(), 0);
  double t_lead = 0;
  const double t_alloc = now_s() - t0;
  long long upd = 0;
  // The TAIL rows on the host pool (round 4).  Row k >= n1 of the up-looking factorisation needs the leading factor (complete
  // after row n1 - 1) and, for its Schur-complement entries, the L21 entries of the tail rows before it.  Two parallel phases with
  // the serial arithmetic, entry by entry:
  //   A  every tail row on its own: reach in the elimination tree, the sparse solve against the LEADING rows of the touched columns
  //      -> its row of L21 (column i, L(k, i), the solve's y_i) and what column k of C leaves in the tail positions;
  //   -- the rows of L21 are appended to the columns in row order (sequential, one pass) --
  //   B  every tail row on its own: its Schur row, C's entries first, then  -= L(j, i) y_i  over the touched columns in the
  //      row's pattern order and each column's tail entries j < k -- the order of the serial loop.
  // PlanarHand_N=10 (m = 483 707, 32 768 tail rows, 28.5 G updates): 11 of the 12 s of the numeric phase were these rows.
  const bool par_tail = tail_k >= 512 && host_pool().size() > 1;
  const int k_serial_end = par_tail ? n1 : m;
  for (int k = 0; k < k_serial_end; ++k) {
    if (k == n1) t_lead = now_s() - t0;
    int top = m;
    flag[k] = k;
    for (int64_t p = Cp[k]; p < Cp[k + 1]; ++p) {
      int i = Ci[p];
      Y[i] += Cx[p];
      int len = 0;
      while (i < k && flag[i] != k) { pattern[len++] = i; flag[i] = k; i = parent[i]; }
      while (len > 0) pattern[--top] = pattern[--len];
    }
    double dk = Y[k];
    Y[k] = 0.0;
    const int top0 = top;
    for (; top < m; ++top) {
      int i = pattern[top];
      if (i >= n1) continue;                      // tail column: its Y entry becomes a Schur-complement entry below
      double yi = Y[i];
      Y[i] = 0.0;
      int64_t p2 = f->Lp[i] + Lnz[i];
      const int* li = f->Li.data();
      const double* lx = f->Lx.data();
      for (int64_t p = f->Lp[i]; p < p2; ++p) Y[li[p]] -= lx[p] * yi;
      upd += p2 - f->Lp[i];
      double lki = yi / f->D[i];
      dk -= lki * yi;
      f->Li[p2] = k;
      f->Lx[p2] = lki;
      Lnz[i]++;
    }
    if (k >= n1) {
      // row k of the Schur complement B22 - L21 D1 L21^T: what the leading columns left in Y (every tail row j < k
      // they touched is in the reach of row k, so the scan of `pattern` collects all of it)
      for (int t = top0; t < m; ++t) {
        int j = pattern[t];
        if (j >= n1) { f->schur_col.push_back(j - n1); f->schur_val.push_back(Y[j]); Y[j] = 0.0; }
      }
      f->schur_col.push_back(k - n1);
      f->schur_val.push_back(dk);
      f->schur_ptr[(size_t)(k - n1) + 1] = (int64_t)f->schur_col.size();
      f->D[k] = 0.0;
      continue;
    }
    if (dk == 0.0 || !std::isfinite(dk)) {
      set_error("Factorization fails! (zero or non-finite pivot at permuted row %d)", k);
      delete f;
      return CUADMM_ERR_FACTOR;
    }
    f->D[k] = dk;
  }
  // (the working set of the two parallel phases -- a second copy of L21 by rows, per-thread m-sized work vectors, the Schur rows --
  // is allocated inside them: an allocation that fails, on this thread or on a pool worker, ends the factorisation with an error
  // code instead of std::terminate)
  std::atomic<bool> tail_oom{false};
  if (par_tail) try {
    t_lead = now_s() - t0;
    struct TailRow { std::vector<int> col; std::vector<double> l, y; std::vector<int> tj; std::vector<double> tc; double dk = 0; };
    std::vector<TailRow> rows((size_t)tail_k);
    std::vector<int> lead_cnt(Lnz.begin(), Lnz.begin() + n1);          // leading entries of every leading column
    const int T = host_pool().size();
    std::atomic<int> next{0};
    // ---- phase A
    host_pool().run([&](int) {
     try {
      std::vector<double> Yt((size_t)m, 0.0);
      std::vector<int> flg((size_t)m, -1), pat((size_t)m);
      for (;;) {
        const int r0 = next.fetch_add(16);
        if (r0 >= tail_k || tail_oom.load(std::memory_order_relaxed)) break;
        for (int r = r0; r < std::min(tail_k, r0 + 16); ++r) {
          const int k = n1 + r;
          TailRow& R = rows[(size_t)r];
          int top = m;
          flg[k] = k;
          for (int64_t p = Cp[k]; p < Cp[k + 1]; ++p) {
            int i = Ci[p];
            Yt[i] += Cx[p];
            int len = 0;
            while (i < k && flg[i] != k) { pat[len++] = i; flg[i] = k; i = parent[i]; }
            while (len > 0) pat[--top] = pat[--len];
          }
          double dk = Yt[k];
          Yt[k] = 0.0;
          for (int t = top; t < m; ++t) {
            const int i = pat[t];
            if (i >= n1) { R.tj.push_back(i); R.tc.push_back(Yt[i]); Yt[i] = 0.0; continue; }    // C's entry of a tail position
            const double yi = Yt[i];
            Yt[i] = 0.0;
            const int64_t pb = f->Lp[i], pe = pb + lead_cnt[i];
            const int* li = f->Li.data();
            const double* lx = f->Lx.data();
            for (int64_t p = pb; p < pe; ++p) Yt[li[p]] -= lx[p] * yi;
            const double lki = yi / f->D[i];
            dk -= lki * yi;
            R.col.push_back(i); R.l.push_back(lki); R.y.push_back(yi);
          }
          R.dk = dk;
        }
      }
     } catch (const std::bad_alloc&) { tail_oom.store(true); }
    });
    (void)T;
    if (tail_oom.load()) throw std::bad_alloc();
    // ---- the rows of L21 into the columns, row order (what the serial loop appends row by row)
    for (int r = 0; r < tail_k; ++r) {
      const TailRow& R = rows[(size_t)r];
      for (size_t q = 0; q < R.col.size(); ++q) {
        const int i = R.col[q];
        const int64_t p2 = f->Lp[i] + Lnz[i];
        f->Li[p2] = n1 + r;
        f->Lx[p2] = R.l[q];
        Lnz[i]++;
      }
    }
    // ---- phase B
    std::vector<std::vector<int>> scol((size_t)tail_k);
    std::vector<std::vector<double>> sval((size_t)tail_k);
    next.store(0);
    host_pool().run([&](int) {
     try {
      std::vector<double> Y2((size_t)tail_k, 0.0);
      for (;;) {
        const int r0 = next.fetch_add(8);
        if (r0 >= tail_k || tail_oom.load(std::memory_order_relaxed)) break;
        for (int r = r0; r < std::min(tail_k, r0 + 8); ++r) {
          const int k = n1 + r;
          const TailRow& R = rows[(size_t)r];
          for (size_t q = 0; q < R.tj.size(); ++q) Y2[(size_t)(R.tj[q] - n1)] = R.tc[q];
          const int* li = f->Li.data();
          const double* lx = f->Lx.data();
          for (size_t q = 0; q < R.col.size(); ++q) {
            const int i = R.col[q];
            const double yi = R.y[q];
            const int64_t pe = f->Lp[i] + Lnz[i];
            for (int64_t p = f->Lp[i] + lead_cnt[i]; p < pe && li[p] < k; ++p) Y2[(size_t)(li[p] - n1)] -= lx[p] * yi;
          }
          auto& sc = scol[(size_t)r];
          auto& sv = sval[(size_t)r];
          sc.reserve(R.tj.size() + 1); sv.reserve(R.tj.size() + 1);
          for (size_t q = 0; q < R.tj.size(); ++q) { const int j = R.tj[q] - n1; sc.push_back(j); sv.push_back(Y2[(size_t)j]); Y2[(size_t)j] = 0.0; }
          sc.push_back(r); sv.push_back(R.dk);
        }
      }
     } catch (const std::bad_alloc&) { tail_oom.store(true); }
    });
    if (tail_oom.load()) throw std::bad_alloc();
    for (int r = 0; r < tail_k; ++r) {
      f->schur_col.insert(f->schur_col.end(), scol[(size_t)r].begin(), scol[(size_t)r].end());
      f->schur_val.insert(f->schur_val.end(), sval[(size_t)r].begin(), sval[(size_t)r].end());
      f->schur_ptr[(size_t)r + 1] = (int64_t)f->schur_col.size();
      std::vector<int>().swap(scol[(size_t)r]); std::vector<double>().swap(sval[(size_t)r]);
      f->D[n1 + r] = 0.0;
    }
  } catch (const std::bad_alloc&) {
    set_error("aat_create: the working set of the %d tail rows of the factor (%lld nonzeros) does not fit in host memory", tail_k, (long long)f->Lp[m]);
    delete f;
    return CUADMM_ERR_FACTOR;
  }
  for (int j = 0; j < n1; ++j) if (f->Lp[j + 1] > f->Lp[j]) f->nzcols.push_back(j);
  if (tail_k > 0) {
    f->lead_mid.resize((size_t)n1);
    for (int j = 0; j < n1; ++j)
      f->lead_mid[j] = (int64_t)(std::lower_bound(f->Li.begin() + f->Lp[j], f->Li.begin() + f->Lp[j + 1], n1) - f->Li.begin());
  }
  // split factor with a large forest of leading trees: chunks for the threaded leading sweeps
  if (tail_k > 0 && n1 >= 20000 && f->Lp[n1] >= 300000) {
    std::vector<int> root(n1);
    for (int j = n1 - 1; j >= 0; --j) root[j] = (parent[j] < 0 || parent[j] >= n1) ? j : root[parent[j]];   // parent[j] > j
    std::vector<int64_t> weight(n1, 0);
    for (int j = 0; j < n1; ++j) weight[root[j]] += 1 + (f->Lp[j + 1] - f->Lp[j]);
    int64_t total = 0, biggest = 0;
    for (int j = 0; j < n1; ++j) if (root[j] == j) { total += weight[j]; biggest = std::max(biggest, weight[j]); }
    constexpr int T = 16;
    if (biggest * 4 < total) {
      // longest-processing-time assignment of the trees (heaviest first onto the lightest chunk): deterministic
      std::vector<int> roots;
      for (int j = 0; j < n1; ++j) if (root[j] == j) roots.push_back(j);
      std::stable_sort(roots.begin(), roots.end(), [&](int a, int b) { return weight[a] > weight[b]; });
      std::vector<int64_t> load(T, 0);
      std::vector<int> chunk_of_root(n1, -1);
      for (int r : roots) {
        int best = 0;
        for (int c = 1; c < T; ++c) if (load[c] < load[best]) best = c;
        chunk_of_root[r] = best; load[best] += weight[r];
      }
      f->lead_chunks.assign(T, {});
      for (int j = 0; j < n1; ++j) if (f->Lp[j + 1] > f->Lp[j]) f->lead_chunks[chunk_of_root[root[j]]].push_back(j);
      f->lead_acc.assign((size_t)T * (size_t)tail_k, 0.0);
    }
  }
  // chunks of independent etree subtrees for the threaded solve (whole factor on the host, large m, many subtrees)
  if (tail_k == 0 && m >= 20000) {
    std::vector<int> root(m);
    for (int j = m - 1; j >= 0; --j) root[j] = parent[j] < 0 ? j : root[parent[j]];   // parent[j] > j
    std::vector<int64_t> weight(m, 0);                                                 // per root: columns + nonzeros
    for (int j = 0; j < m; ++j) weight[root[j]] += 1 + (f->Lp[j + 1] - f->Lp[j]);
    int64_t total = 0, biggest = 0;
    for (int j = 0; j < m; ++j) if (root[j] == j) { total += weight[j]; biggest = std::max(biggest, weight[j]); }
    const int T = 16;
    if (biggest * 4 < total) {                      // no dominating subtree: worth splitting
      std::vector<int> chunk_of_root(m, -1);
      int64_t acc = 0;
      for (int j = 0; j < m; ++j)                   // roots in ascending order of the root column
        if (root[j] == j) { chunk_of_root[j] = (int)std::min<int64_t>(T - 1, acc * T / total); acc += weight[j]; }
      f->chunks.assign(T, {});
      for (int j = 0; j < m; ++j) f->chunks[chunk_of_root[root[j]]].push_back(j);
    }
  }
  f->factor_s = now_s() - t0;
  if (getenv("CUADMM_AAT_TIMING")) fprintf(stderr, "[aat] analyze %.3fs numeric %.3fs (alloc %.3fs, leading rows until %.3fs, %lld updates) nnz(L) %lld tail %d\n", f->analyze_s, f->factor_s, t_alloc, t_lead, upd, (long long)f->Lp[m], tail_k);
  *out = f;
  return CUADMM_OK;
}

}  // namespace

extern "C" {

const int* cuadmm_aat_perm(const cuadmm_aat* f) { return f ? f->perm.data() : nullptr; }
int64_t cuadmm_aat_factor_nnz(const cuadmm_aat* f) { return f ? f->Lp[f->m] : 0; }
const int64_t* cuadmm_aat_factor_colptr(const cuadmm_aat* f) { return f ? f->Lp.data() : nullptr; }

int cuadmm_aat_tail_k(const cuadmm_aat* f) { return f ? f->tail_k : 0; }
int cuadmm_aat_tail_tops(const cuadmm_aat* f) { return f ? f->plan_tops : 0; }
void cuadmm_aat_plan_allow_tops(int allow) { g_plan_allow_tops = allow != 0; }

int cuadmm_aat_factor_arrays(const cuadmm_aat* f, const int64_t** Lp, const int** Li, const double** Lx, const double** D) {
  if (!f || !Lp || !Li || !Lx || !D) { set_error("aat_factor_arrays: null argument"); return CUADMM_ERR_INVALID; }
  *Lp = f->Lp.data(); *Li = f->Li.data(); *Lx = f->Lx.data(); *D = f->D.data();
  return CUADMM_OK;
}

// The elimination forest of the factor: parent(j) = smallest row index of column j of L.  Independent trees never exchange
// data in a solve, so many small trees (block-diagonal A A^T: one tree per group of coupled constraints) can be solved by
// one GPU thread each (engine.hip: forest_solve_kernel) with the arithmetic order of the serial host sweeps.
int cuadmm_aat_forest(cuadmm_aat* f, int* n_trees, int* max_cols, const int** tree_ptr, const int** tree_cols) {
  if (!f || !n_trees || !max_cols || !tree_ptr || !tree_cols) { set_error("aat_forest: null argument"); return CUADMM_ERR_INVALID; }
  if (f->tail_k > 0) { set_error("aat_forest: the factor is split"); return CUADMM_ERR_INVALID; }
  const int m = f->m;
  if (f->forest_ptr.empty()) {
    std::vector<int> root((size_t)m), parent((size_t)m, -1);
    for (int j = 0; j < m; ++j) {
      int pmin = -1;
      for (int64_t p = f->Lp[j]; p < f->Lp[j + 1]; ++p) if (pmin < 0 || f->Li[p] < pmin) pmin = f->Li[p];
      parent[j] = pmin;
    }
    for (int j = m - 1; j >= 0; --j) root[j] = parent[j] < 0 ? j : root[parent[j]];   // parent(j) > j
    std::vector<int> tree_of_root((size_t)m, -1), count;
    for (int j = 0; j < m; ++j) {
      if (tree_of_root[root[j]] < 0) { tree_of_root[root[j]] = (int)count.size(); count.push_back(0); }
      count[tree_of_root[root[j]]]++;
    }
    f->forest_ptr.assign(count.size() + 1, 0);
    for (size_t t = 0; t < count.size(); ++t) { f->forest_ptr[t + 1] = f->forest_ptr[t] + count[t]; f->forest_max = std::max(f->forest_max, count[t]); }
    f->forest_cols.resize((size_t)m);
    std::vector<int> fill(f->forest_ptr.begin(), f->forest_ptr.end() - 1);
    for (int j = 0; j < m; ++j) f->forest_cols[fill[tree_of_root[root[j]]]++] = j;     // ascending inside every tree
  }
  *n_trees = (int)f->forest_ptr.size() - 1;
  *max_cols = f->forest_max;
  *tree_ptr = f->forest_ptr.data();
  *tree_cols = f->forest_cols.data();
  return CUADMM_OK;
}
int cuadmm_aat_tail_schur(const cuadmm_aat* f, const int64_t** row_ptr, const int** col, const double** val) {
  if (!f || !row_ptr || !col || !val) { set_error("aat_tail_schur: null argument"); return CUADMM_ERR_INVALID; }
  if (f->tail_k == 0 || f->schur_ptr.empty()) { set_error("aat_tail_schur: no Schur complement (factor not split, or released)"); return CUADMM_ERR_INVALID; }
  *row_ptr = f->schur_ptr.data(); *col = f->schur_col.data(); *val = f->schur_val.data();
  return CUADMM_OK;
}
void cuadmm_aat_tail_schur_release(cuadmm_aat* f) {
  if (!f) return;
  std::vector<int64_t>().swap(f->schur_ptr); std::vector<int>().swap(f->schur_col); std::vector<double>().swap(f->schur_val);
}

int cuadmm_aat_solve_permuted(const cuadmm_aat* f, const double* rhs, double* x) {
  if (f && f->m == 0) return CUADMM_OK;   // a rank without constraints (owned-constraints sharding)
  if (!f || !rhs || !x) { set_error("aat_solve: null argument"); return CUADMM_ERR_INVALID; }
  if (f->tail_k > 0) { set_error("aat_solve: the factor is split (its last %d columns live on the GPU); use the leading sweeps", f->tail_k); return CUADMM_ERR_INVALID; }
  const int m = f->m;
  if (x != rhs) std::memcpy(x, rhs, sizeof(double) * (size_t)m);
  const int64_t* Lp = f->Lp.data();
  const int* Li = f->Li.data();
  const double* Lx = f->Lx.data();
  const double* D = f->D.data();
  if (!f->chunks.empty() && host_pool().size() > 1) {
    // every chunk is a union of whole etree subtrees: its three sweeps touch only its own entries of x
    const int nchunk = (int)f->chunks.size(), T = host_pool().size();
    host_pool().run([&](int t) {
      for (int c = t; c < nchunk; c += T) {
        const std::vector<int>& cols = f->chunks[c];
        for (int j : cols) {
          const double xj = x[j];
          if (xj != 0.0)
            for (int64_t p = Lp[j]; p < Lp[j + 1]; ++p) x[Li[p]] -= Lx[p] * xj;
        }
        for (size_t q = cols.size(); q-- > 0;) {   // D^-1 folded into the backward sweep (same arithmetic per entry)
          const int j = cols[q];
          double s = x[j] / D[j];
          for (int64_t p = Lp[j]; p < Lp[j + 1]; ++p) s -= Lx[p] * x[Li[p]];
          x[j] = s;
        }
      }
    });
    return CUADMM_OK;
  }
  // only columns with sub-diagonal entries take part in the sweeps; D^-1 is one streaming pass in between (per entry
  // the arithmetic is that of the textbook three-pass solve: updates, division, subtractions)
  const int* nz = f->nzcols.data();
  const size_t nnzc = f->nzcols.size();
  for (size_t q = 0; q < nnzc; ++q) {  // L z = b
    const int j = nz[q];
    const double xj = x[j];
    if (xj != 0.0)
      for (int64_t p = Lp[j]; p < Lp[j + 1]; ++p) x[Li[p]] -= Lx[p] * xj;
  }
  for (int j = 0; j < m; ++j) x[j] /= D[j];
  for (size_t q = nnzc; q-- > 0;) {  // L^T x = z
    const int j = nz[q];
    double s = x[j];
    for (int64_t p = Lp[j]; p < Lp[j + 1]; ++p) s -= Lx[p] * x[Li[p]];
    x[j] = s;
  }
  return CUADMM_OK;
}

// ---------------------------------------------------------------------------------------
// Dense-tail split.  With a fill-reducing ordering the last k columns of L form an (almost) dense triangle that
// holds most of nnz(L) (PlanarHand_N=1: last 12 000 of 66 008 columns = 92 % of 13.5 M).  The engine keeps
// inv(L22) in HBM and runs that part of both triangular solves as two GEMVs on the GPU (tail_solve.hip); the host
// keeps the sparse leading columns:
//   forward  : columns j < m-k  (updates rows of both parts),           host
//   tail     : x2 = L22^-T D2^-1 L22^-1 z2,                             GPU
//   backward : columns j < m-k  (reads x2),                             host
// ---------------------------------------------------------------------------------------
// Cost model: host 1.2 ns per nonzero of the leading columns (both sweeps, measured), GPU 60 us + k^2 * 8 B at 4 TB/s.
int cuadmm_aat_tail_plan(const cuadmm_aat* f, int max_k) { return f ? plan_tail(f->Lp.data(), f->m, max_k) : 0; }

// dense (k x ld, row-major, unit lower triangular, ld >= k) copy of the trailing k x k block of L, and its D
int cuadmm_aat_tail_dense(const cuadmm_aat* f, int k, double* L22, int64_t ld, double* D2) {
  if (!f || !L22 || !D2 || k < 1 || k > f->m || ld < k) { set_error("aat_tail_dense: bad arguments"); return CUADMM_ERR_INVALID; }
  if (f->tail_k > 0) { set_error("aat_tail_dense: the factor is split, its tail was never factored on the host"); return CUADMM_ERR_INVALID; }
  const int m = f->m, n1 = m - k;
  for (int i = 0; i < k; ++i) {
    std::memset(L22 + (size_t)i * ld, 0, sizeof(double) * (size_t)ld);
    L22[(size_t)i * ld + i] = 1.0;
    D2[i] = f->D[n1 + i];
  }
  for (int j = n1; j < m; ++j)
    for (int64_t p = f->Lp[j]; p < f->Lp[j + 1]; ++p) L22[(size_t)(f->Li[p] - n1) * ld + (j - n1)] = f->Lx[p];
  return CUADMM_OK;
}

// forward sweep over the leading m-k columns; x[m-k..] holds z2 on return (D1^-1 is applied by the backward sweep)
int cuadmm_aat_solve_leading_forward(const cuadmm_aat* f, int k, double* x) {
  if (!f || !x || k < 0 || k > f->m || (f->tail_k > 0 && k != f->tail_k)) { set_error("aat_solve_leading_forward: bad arguments (a split factor takes its own tail size)"); return CUADMM_ERR_INVALID; }
  const int n1 = f->m - k;
  const int64_t* Lp = f->Lp.data();
  const int* Li = f->Li.data();
  const double* Lx = f->Lx.data();
  if (k != f->tail_k && f->tail_k == 0) {   // split requested on a one-piece factor (tests): walk all leading columns
    for (int j = 0; j < n1; ++j) {
      const double xj = x[j];
      if (xj != 0.0)
        for (int64_t p = Lp[j]; p < Lp[j + 1]; ++p) x[Li[p]] -= Lx[p] * xj;
    }
    return CUADMM_OK;
  }
  if (!f->lead_chunks.empty() && host_pool().size() > 1) {
    const int nchunk = (int)f->lead_chunks.size(), T = host_pool().size();
    double* acc_all = f->lead_acc.data();
    host_pool().run([&](int t) {
      for (int c = t; c < nchunk; c += T) {
        double* acc = acc_all + (size_t)c * (size_t)k - n1;       // acc[i] for tail rows i >= n1
        std::memset(acc + n1, 0, sizeof(double) * (size_t)k);
        for (int j : f->lead_chunks[c]) {
          const double xj = x[j];
          if (xj != 0.0)
            for (int64_t p = Lp[j]; p < Lp[j + 1]; ++p) {
              const int i = Li[p];
              if (i < n1) x[i] -= Lx[p] * xj; else acc[i] -= Lx[p] * xj;
            }
        }
      }
    });
    for (int c = 0; c < nchunk; ++c) {                              // chunk order: the same bits for every thread count
      const double* acc = acc_all + (size_t)c * (size_t)k;
      for (int i = 0; i < k; ++i) x[n1 + i] += acc[i];
    }
    return CUADMM_OK;
  }
  for (int j : f->nzcols) {
    const double xj = x[j];
    if (xj != 0.0)
      for (int64_t p = Lp[j]; p < Lp[j + 1]; ++p) x[Li[p]] -= Lx[p] * xj;
  }
  return CUADMM_OK;   // the D1^-1 scaling of the leading part is applied by the backward sweep
}

// D1^-1 and the backward sweep over the leading m-k columns; x[m-k..] must hold the solved tail x2
int cuadmm_aat_solve_leading_backward(const cuadmm_aat* f, int k, double* x) {
  if (!f || !x || k < 0 || k > f->m || (f->tail_k > 0 && k != f->tail_k)) { set_error("aat_solve_leading_backward: bad arguments (a split factor takes its own tail size)"); return CUADMM_ERR_INVALID; }
  const int n1 = f->m - k;
  const int64_t* Lp = f->Lp.data();
  const int* Li = f->Li.data();
  const double* Lx = f->Lx.data();
  const double* D = f->D.data();
  for (int j = 0; j < n1; ++j) x[j] /= D[j];
  if (k != f->tail_k && f->tail_k == 0) {
    for (int j = n1 - 1; j >= 0; --j) {
      double s = x[j];
      for (int64_t p = Lp[j]; p < Lp[j + 1]; ++p) s -= Lx[p] * x[Li[p]];
      x[j] = s;
    }
    return CUADMM_OK;
  }
  if (!f->lead_chunks.empty() && host_pool().size() > 1) {       // a chunk reads the solved tail and its own trees' entries
    const int nchunk = (int)f->lead_chunks.size(), T = host_pool().size();
    host_pool().run([&](int t) {
      for (int c = t; c < nchunk; c += T) {
        const std::vector<int>& cols = f->lead_chunks[c];
        for (size_t q = cols.size(); q-- > 0;) {
          const int j = cols[q];
          double s = x[j];
          for (int64_t p = Lp[j]; p < Lp[j + 1]; ++p) s -= Lx[p] * x[Li[p]];
          x[j] = s;
        }
      }
    });
    return CUADMM_OK;
  }
  for (size_t q = f->nzcols.size(); q-- > 0;) {
    const int j = f->nzcols[q];
    double s = x[j];
    for (int64_t p = Lp[j]; p < Lp[j + 1]; ++p) s -= Lx[p] * x[Li[p]];
    x[j] = s;
  }
  return CUADMM_OK;
}

// The same two sweeps restricted to L11 (rows and columns < m - k) for a split factor whose L21 lives on the GPU (engine: lead_solve.hip,
// hybrid mode -- PlanarHand_N=10: 70 % of the leading nonzeros are in the tail rows):
//   forward11 : x1 <- L11^-1 x1; x[m-k..] is NOT touched (the GPU forms z2 = x2 - L21 x1)
//   backward11: x1 <- L11^-T (D1^-1 x1 - w) with w = L21^T x2 from the GPU (m - k doubles); x[m-k..] is not read
int cuadmm_aat_solve_leading_forward11(const cuadmm_aat* f, int k, double* x) {
  if (!f || !x || f->tail_k <= 0 || k != f->tail_k || f->lead_mid.empty()) { set_error("aat_solve_leading_forward11: a split factor and its own tail size"); return CUADMM_ERR_INVALID; }
  const int64_t* Lp = f->Lp.data();
  const int64_t* Lm = f->lead_mid.data();
  const int* Li = f->Li.data();
  const double* Lx = f->Lx.data();
  auto col = [&](int j) {
    const double xj = x[j];
    if (xj != 0.0)
      for (int64_t p = Lp[j]; p < Lm[j]; ++p) x[Li[p]] -= Lx[p] * xj;
  };
  if (!f->lead_chunks.empty() && host_pool().size() > 1) {       // trees of different chunks share no leading row
    const int nchunk = (int)f->lead_chunks.size(), T = host_pool().size();
    host_pool().run([&](int t) {
      for (int c = t; c < nchunk; c += T)
        for (int j : f->lead_chunks[c]) col(j);
    });
    return CUADMM_OK;
  }
  for (int j : f->nzcols) col(j);
  return CUADMM_OK;
}

int cuadmm_aat_solve_leading_backward11(const cuadmm_aat* f, int k, double* x, const double* w) {
  if (!f || !x || !w || f->tail_k <= 0 || k != f->tail_k || f->lead_mid.empty()) { set_error("aat_solve_leading_backward11: a split factor and its own tail size"); return CUADMM_ERR_INVALID; }
  const int n1 = f->m - k;
  const int64_t* Lp = f->Lp.data();
  const int64_t* Lm = f->lead_mid.data();
  const int* Li = f->Li.data();
  const double* Lx = f->Lx.data();
  const double* D = f->D.data();
  auto col = [&](int j) {
    double s = x[j] / D[j] - w[j];
    for (int64_t p = Lp[j]; p < Lm[j]; ++p) s -= Lx[p] * x[Li[p]];
    x[j] = s;
  };
  if (!f->lead_chunks.empty() && host_pool().size() > 1) {
    const int nchunk = (int)f->lead_chunks.size(), T = host_pool().size();
    for (int j = 0; j < n1; ++j) if (Lp[j + 1] == Lp[j]) x[j] = x[j] / D[j] - w[j];        // columns without entries are in no chunk; others read them
    host_pool().run([&](int t) {
      for (int c = t; c < nchunk; c += T) {
        const std::vector<int>& cols = f->lead_chunks[c];
        for (size_t q = cols.size(); q-- > 0;) col(cols[q]);
      }
    });
    return CUADMM_OK;
  }
  for (int j = n1 - 1; j >= 0; --j) col(j);
  return CUADMM_OK;
}

void cuadmm_aat_free(cuadmm_aat* f) { delete f; }

// how many ranks share this node's CPUs (before the pool's first use; later calls are ignored)
void cuadmm_host_pool_hint(int ranks_on_node) { if (ranks_on_node > 0) g_pool_ranks_hint.store(ranks_on_node); }
int cuadmm_host_pool_threads(void) { return host_pool().size(); }

// fn(chunk, ctx) for chunk = 0..nchunks-1 on the host pool (CUADMM_HOST_THREADS); chunks are handed out statically
// (chunk c runs on thread c mod T), so anything reduced per chunk and combined in chunk order is reproducible.
void cuadmm_host_parallel_for(int nchunks, void (*fn)(int, void*), void* ctx) {
  if (nchunks <= 0 || !fn) return;
  HostPool& pool = host_pool();
  const int T = pool.size();
  if (T == 1 || nchunks == 1) { for (int c = 0; c < nchunks; ++c) fn(c, ctx); return; }
  pool.run([&](int t) { for (int c = t; c < nchunks; c += T) fn(c, ctx); });
}

}  // extern "C"
