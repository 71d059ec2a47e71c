// SDPDuoSolver's "N devices from ONE process" mode (reference src/duo_solver.cu:487-577: the moment matrices are spread over
// device_num_requested GPUs by 1 + N host threads and copied between them peer to peer, src/utils/check_gpus.cu:29-43).
//
// Here a GPU is a rank of the block-sharded engine.  cuadmm_duo_init(..., device_num_requested = N) on a
// handle whose world is 1 builds a GROUP: the caller's handle becomes rank 0, N - 1 child engines are created with the same
// options (the option log is replayed) on devices 0 .. N-1 -- or all on the caller's device with option "duo_share_device" = 1 --
// and every collective of the engine goes through an in-process all-reduce.  Two exchanges, chosen when the group is created
// (option "duo_exchange": -1 = choose, 0 = host, 1 = device; cuadmm_get_group_info names the one in use):
//   * DEVICE (the default whenever every rank's device can read every other's memory -- one device shared by all ranks, or
//     peer access over xGMI as the reference's P2P copies, check_gpus.cu:29-43, duo_solver.cu:598-606): each rank copies its
//     buffer into a device staging buffer of its own (two of them, alternating), the host threads meet at ONE barrier, and each
//     rank's stream runs a kernel that adds the N staging buffers IN RANK ORDER straight out of the peers' memory into its own
//     buffer.  Nothing crosses PCIe, no host arithmetic; the sums are bit-identical on every rank, as the replicated solve requires.
//     (The alternation makes a second barrier unnecessary: a rank re-writes a staging buffer two collectives later, and every
//     reader has synchronised its stream -- at the start of the collective in between -- before anybody gets there.)
//   * HOST (devices without peer access): page-locked portable staging buffers, two barriers, every rank adds the N buffers in
//     rank order on the host and copies the sum back: O(N^2 m) host work and two PCIe hops -- the fallback, not the design.
// The exchanged vector is [A X | sums | A (S - C)], 2m+2 doubles (or 4 scalars when constraints are owned).
//
// init and solve of the ranks run on N host threads (the caller's thread is rank 0); the getters of the caller's handle gather
// the shards.  Failure protocol: a rank that fails -- an error code, a C++ exception (std::bad_alloc in a large factor), or a
// return from its call while the others are still inside a collective -- raises the group's abort flag; every barrier wakes up
// and returns an error (CUADMM_ERR_COMM in the engine), no rank is left waiting, no std::thread is destroyed joinable.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <functional>
#include <condition_variable>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "common.h"
#include "cuadmm_amd.h"
#include "duo_group.h"

namespace cuadmm {

struct DuoGroup;
struct DuoRank {
  DuoGroup* g = nullptr; int rank = 0; int device = 0;
  double* stage = nullptr; double* sum = nullptr; size_t cap = 0;       // host exchange: page-locked, portable
  double* dstage[2] = {nullptr, nullptr}; size_t dcap = 0;              // device exchange: on this rank's device
  long long calls = 0;         // collectives of the CURRENT duo_group_run call (zeroed for every rank when a call starts: the staging slot
                               // and the published index derive from it, so the ranks re-agree on it after a call that failed half way)
  long long inj_calls = 0;     // collectives since the test hook was armed (duo_group_inject)
};

struct DuoGroup {
  int world = 1;
  std::vector<cuadmm_solver*> child;       // [0] = the caller's handle (not owned), [r >= 1] owned
  std::vector<DuoRank> ranks;
  std::mutex mu;
  std::condition_variable cv;
  int arrived = 0;
  int left = 0;                            // ranks that have returned from the current duo_group_run call
  long long generation = 0;
  std::atomic<bool> abort{false};
  // what each rank brought to a collective (must agree) and, device exchange, where it staged it -- PUBLISHED per collective parity:
  // with one barrier per collective a rank may already be publishing for collective k + 1 while a slower one still reads what was
  // published for k (it cannot get further: barrier k + 1 needs everybody), so two sets suffice
  std::vector<size_t> counts;              // [parity * world + rank]
  std::vector<long long> index;            // [parity * world + rank]: which collective of the call the entry belongs to
  std::vector<const double*> pub;          // [parity * world + rank]
  long long n_allreduce = 0;
  bool device_exchange = false;
  int distinct_devices = 1;
  std::vector<std::pair<int, double*>> retired;   // outgrown device staging buffers (device, pointer): freed with the group
  long long inject = 0;                    // test hook (option "duo_inject_fail"): rank r fails its k-th collective (r * 1e6 + k),
                                           // throws instead with the sign flipped

  // returns false when the group was aborted (a rank failed), when a rank has LEFT the call the others are still communicating in
  // (it would never arrive), or when the ranks disagree on the length
  bool barrier() {
    std::unique_lock<std::mutex> lk(mu);
    if (abort.load()) return false;
    if (left > 0) { abort.store(true); cv.notify_all(); return false; }
    const long long gen = generation;
    if (++arrived == world) { arrived = 0; ++generation; cv.notify_all(); return !abort.load(); }
    cv.wait(lk, [&] { return generation != gen || abort.load(); });
    return generation != gen && !abort.load();
  }
  void raise_abort() {
    { std::lock_guard<std::mutex> lk(mu); abort.store(true); }
    cv.notify_all();
  }
  // a rank is back from fn: if somebody is waiting for it at a barrier, that barrier can never complete
  void rank_left() {
    std::lock_guard<std::mutex> lk(mu);
    ++left;
    if (arrived > 0) { abort.store(true); cv.notify_all(); }
  }
  void publish(int parity, int r, size_t c, long long idx, const double* where) {   // read by the others after the barrier
    std::lock_guard<std::mutex> lk(mu);
    counts[(size_t)parity * world + r] = c;
    index[(size_t)parity * world + r] = idx;
    pub[(size_t)parity * world + r] = where;
  }
  // every rank brought the same length to the same collective of this call: a stale entry of an earlier collective (a rank that
  // failed and came back with another count of calls) carries another index even when its length matches
  bool counts_agree(int parity, size_t c, long long idx) {
    std::lock_guard<std::mutex> lk(mu);
    for (int r = 0; r < world; ++r)
      if (counts[(size_t)parity * world + r] != c || index[(size_t)parity * world + r] != idx) return false;
    return true;
  }
  const double* published(int parity, int r) { std::lock_guard<std::mutex> lk(mu); return pub[(size_t)parity * world + r]; }
};

// buf[i] = sum over ranks (in rank order) of stage_r[i]: every rank runs it on its own stream, reading its peers' memory
struct DuoPeers { const double* p[16]; };
__global__ void duo_sum_kernel(DuoPeers peers, int world, double* __restrict__ buf, size_t count) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (size_t)gridDim.x * blockDim.x) {
    double s = peers.p[0][i];
    for (int r = 1; r < world; ++r) s += peers.p[r][i];
    buf[i] = s;
  }
}

// test hook: inject = +-(rank * 1e6 + k) fails rank's k-th collective since the hook was armed (negative: throws instead);
// + 5e8: the rank fails IN FRONT of that collective, without counting it -- a failure between two collectives, which leaves the ranks'
// counters apart
constexpr long long kInjectBetween = 500000000;
static bool duo_injected_failure(DuoRank* me, bool before_counting) {
  const long long inj = me->g->inject;
  if (inj == 0) return false;
  long long a = inj < 0 ? -inj : inj;
  const bool between = a >= kInjectBetween;
  if (between) a -= kInjectBetween;
  if (between != before_counting) return false;
  if (a / 1000000 != me->rank || a % 1000000 != me->inj_calls + (before_counting ? 1 : 0)) return false;
  if (inj < 0) throw std::bad_alloc();
  return true;
}

static int duo_allreduce_device(DuoRank* me, double* buf, size_t count, hipStream_t st) {
  DuoGroup* g = me->g;
  if (me->dcap < count) {
    // grown by every rank at the same collective (the lengths agree); the old buffers are no longer read: every rank synchronised
    // its stream when it entered this collective... which is only known after the barrier below, so the old ones are kept until
    // the group is destroyed instead of freed here
    const size_t cap = count + count / 2 + 64;
    double *a = nullptr, *b = nullptr;
    if (hipMalloc(reinterpret_cast<void**>(&a), cap * sizeof(double)) != hipSuccess || hipMalloc(reinterpret_cast<void**>(&b), cap * sizeof(double)) != hipSuccess) {
      g->raise_abort();
      return 1;
    }
    { std::lock_guard<std::mutex> lk(g->mu); g->retired.push_back({me->device, me->dstage[0]}); g->retired.push_back({me->device, me->dstage[1]}); }
    me->dstage[0] = a; me->dstage[1] = b; me->dcap = cap;
  }
  const int slot = (int)(me->calls & 1);
  if (hipMemcpyAsync(me->dstage[slot], buf, count * sizeof(double), hipMemcpyDeviceToDevice, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) {
    g->raise_abort();
    return 1;
  }
  g->publish(slot, me->rank, count, me->calls, me->dstage[slot]);
  if (!g->barrier()) return 1;
  if (!g->counts_agree(slot, count, me->calls)) { g->raise_abort(); return 2; }   // ranks issued different collectives
  DuoPeers peers;
  for (int r = 0; r < g->world; ++r) peers.p[r] = g->published(slot, r);     // published before the barrier by their owners
  const unsigned grid = (unsigned)std::min<size_t>((count + 255) / 256, 1024);
  hipLaunchKernelGGL(duo_sum_kernel, dim3(grid), dim3(256), 0, st, peers, g->world, buf, count);
  if (hipGetLastError() != hipSuccess) { g->raise_abort(); return 1; }
  if (me->rank == 0) ++g->n_allreduce;
  return 0;
}

static int duo_allreduce_host(DuoRank* me, double* buf, size_t count, hipStream_t st) {
  DuoGroup* g = me->g;
  if (me->cap < count) {
    if (me->stage) { (void)hipHostFree(me->stage); (void)hipHostFree(me->sum); me->stage = me->sum = nullptr; me->cap = 0; }
    const size_t cap = count + count / 2 + 64;
    if (hipHostMalloc(reinterpret_cast<void**>(&me->stage), cap * sizeof(double), hipHostMallocPortable) != hipSuccess ||
        hipHostMalloc(reinterpret_cast<void**>(&me->sum), cap * sizeof(double), hipHostMallocPortable) != hipSuccess) {
      g->raise_abort();
      return 1;
    }
    me->cap = cap;
  }
  if (hipMemcpyAsync(me->stage, buf, count * sizeof(double), hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) {
    g->raise_abort();
    return 1;
  }
  g->publish(0, me->rank, count, me->calls, nullptr);                        // (two barriers per collective here: one set is enough)
  if (!g->barrier()) return 1;
  if (!g->counts_agree(0, count, me->calls)) { g->raise_abort(); return 2; }   // ranks issued different collectives
  // every rank forms the same sum in the same order
  const double* s0 = g->ranks[0].stage;
  for (size_t i = 0; i < count; ++i) me->sum[i] = s0[i];
  for (int r = 1; r < g->world; ++r) {
    const double* sr = g->ranks[(size_t)r].stage;
    for (size_t i = 0; i < count; ++i) me->sum[i] += sr[i];
  }
  if (!g->barrier()) return 1;                                               // nobody overwrites a staging buffer that is still being read
  if (hipMemcpyAsync(buf, me->sum, count * sizeof(double), hipMemcpyHostToDevice, st) != hipSuccess) { g->raise_abort(); return 1; }
  if (me->rank == 0) ++g->n_allreduce;
  return 0;
}

static int duo_allreduce_hook(void* user, double* buf, size_t count, void* hip_stream) {
  DuoRank* me = static_cast<DuoRank*>(user);
  hipStream_t st = static_cast<hipStream_t>(hip_stream);
  if (count == 0) return 0;
  if (duo_injected_failure(me, true)) { me->g->raise_abort(); return 1; }
  ++me->calls; ++me->inj_calls;
  if (duo_injected_failure(me, false)) { me->g->raise_abort(); return 1; }
  return me->g->device_exchange ? duo_allreduce_device(me, buf, count, st) : duo_allreduce_host(me, buf, count, st);
}

DuoGroup* duo_group_of(void* p) { return static_cast<DuoGroup*>(p); }

void duo_group_destroy(void* p) {
  DuoGroup* g = duo_group_of(p);
  if (!g) return;
  for (size_t r = 1; r < g->child.size(); ++r) cuadmm_destroy(g->child[r]);
  int cur = 0;
  const bool have_cur = hipGetDevice(&cur) == hipSuccess;
  for (auto& rk : g->ranks) {
    if (rk.stage) (void)hipHostFree(rk.stage);
    if (rk.sum) (void)hipHostFree(rk.sum);
    for (double* d : rk.dstage) if (d) { (void)hipSetDevice(rk.device); (void)hipFree(d); }
  }
  for (auto& kv : g->retired) if (kv.second) { (void)hipSetDevice(kv.first); (void)hipFree(kv.second); }
  if (have_cur) (void)hipSetDevice(cur);
  delete g;
}

int duo_group_world(void* p) { return p ? duo_group_of(p)->world : 1; }
cuadmm_solver* duo_group_rank(void* p, int r) { DuoGroup* g = duo_group_of(p); return (g && r >= 0 && r < g->world) ? g->child[(size_t)r] : nullptr; }
long long duo_group_allreduces(void* p) { return p ? duo_group_of(p)->n_allreduce : 0; }
int duo_group_exchange(void* p) { return p && duo_group_of(p)->device_exchange ? 1 : 0; }
int duo_group_distinct_devices(void* p) { return p ? duo_group_of(p)->distinct_devices : 1; }
void duo_group_inject(void* p, long long v) {
  if (!p) return;
  DuoGroup* g = duo_group_of(p);
  g->inject = v;
  for (auto& rk : g->ranks) rk.inj_calls = 0;       // the hook counts from the moment it is armed
}

// Builds the group around `parent` (rank 0): children with the parent's options, the hook on every rank.  The caller then runs
// `fn(rank_handle)` on every rank through duo_group_run (init, solve).
int duo_group_create(cuadmm_solver* parent, int world, int parent_device, bool share_device, int exchange,
                     const std::vector<std::pair<std::string, double>>& option_log, void** out) {

  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) { set_error("duo_init: no HIP device"); return CUADMM_ERR_NO_DEVICE; }
  if (!share_device && world > ndev) {
    set_error("duo_init: device_num_requested = %d but %d device(s) are visible (option duo_share_device = 1 runs every engine on device %d)", world, ndev, parent_device);
    return CUADMM_ERR_INVALID;
  }
  if (world > 16) { set_error("duo_init: at most 16 engines per group (%d requested)", world); return CUADMM_ERR_INVALID; }
  DuoGroup* g = new DuoGroup();
  g->world = world;
  g->child.assign((size_t)world, nullptr);
  g->ranks.assign((size_t)world, DuoRank{});
  g->counts.assign(2 * (size_t)world, 0);
  g->index.assign(2 * (size_t)world, -1);
  g->pub.assign(2 * (size_t)world, nullptr);
  g->child[0] = parent;
  for (int r = 0; r < world; ++r) {
    g->ranks[(size_t)r].g = g; g->ranks[(size_t)r].rank = r;
    // check_gpus.cu:29-43 walks devices 0 .. N-1; rank r takes device r (the parent keeps its own)
    g->ranks[(size_t)r].device = (share_device || r == 0) ? parent_device : (r == parent_device ? 0 : r);
  }
  g->distinct_devices = share_device ? 1 : world;
  // the exchange: through device memory when every rank can read every other rank's staging buffer
  bool peers_ok = true;
  if (!share_device && exchange != 0) {
    int cur = 0;
    (void)hipGetDevice(&cur);
    for (int a = 0; a < world && peers_ok; ++a)
      for (int b = 0; b < world && peers_ok; ++b) {
        const int da = g->ranks[(size_t)a].device, db = g->ranks[(size_t)b].device;
        if (da == db) continue;
        int can = 0;
        if (hipDeviceCanAccessPeer(&can, da, db) != hipSuccess || !can) { peers_ok = false; break; }
        if (hipSetDevice(da) != hipSuccess) { peers_ok = false; break; }
        const hipError_t e = hipDeviceEnablePeerAccess(db, 0);
        if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) peers_ok = false;
        (void)hipGetLastError();
      }
    (void)hipSetDevice(cur);
  }
  if (exchange == 1 && !peers_ok) {
    set_error("duo_init: option duo_exchange = 1 (device-side exchange) but the %d devices cannot all access each other's memory", world);
    delete g;
    return CUADMM_ERR_INVALID;
  }
  g->device_exchange = exchange != 0 && peers_ok;
  int rc = CUADMM_OK;
  for (int r = 1; r < world && !rc; ++r) {
    cuadmm_solver* c = nullptr;
    if ((rc = cuadmm_create(&c))) break;
    g->child[(size_t)r] = c;
    for (const auto& kv : option_log) {
      if (kv.first == "device" || kv.first == "rank" || kv.first == "world" || kv.first == "verbose") continue;
      if ((rc = cuadmm_set_option(c, kv.first.c_str(), kv.second))) break;
    }
    if (rc) break;
    const int dev = g->ranks[(size_t)r].device;
    if ((rc = cuadmm_set_option(c, "device", dev)) || (rc = cuadmm_set_option(c, "verbose", 0)) || (rc = cuadmm_set_option(c, "rank", r)) ||
        (rc = cuadmm_set_option(c, "world", world)) || (rc = cuadmm_set_allreduce(c, duo_allreduce_hook, &g->ranks[(size_t)r])))
      break;
  }
  if (!rc) rc = cuadmm_set_allreduce(parent, duo_allreduce_hook, &g->ranks[0]);
  if (rc) { duo_group_destroy(g); return rc; }
  *out = g;
  return CUADMM_OK;
}

// fn on every rank, ranks >= 1 on their own host threads, rank 0 on the caller's; the first error wins (its message is
// re-raised on the caller's thread: set_error is thread-local)
int duo_group_run(void* p, const std::function<int(cuadmm_solver*, int)>& fn) {
  DuoGroup* g = duo_group_of(p);
  const int N = g->world;
  // All threads of the previous call have been joined.  The ranks re-agree on everything the exchange derives from their call
  // counters: after a call that failed half way the counters differ (a rank that failed between two collectives is one behind the
  // rank that entered the next one), and with them the staging slot and the publication set each rank would use.
  // Every call starts with idle devices: the first collective of this call reuses staging slot 1 whatever the previous call's last
  // one used, and after a FAILED call kernels of its last collective may still be reading their peers' staging buffers.
  {
    int cur = 0;
    const bool have_cur = hipGetDevice(&cur) == hipSuccess;
    for (auto& rk : g->ranks) if (hipSetDevice(rk.device) == hipSuccess) (void)hipDeviceSynchronize();
    (void)hipGetLastError();
    if (have_cur) (void)hipSetDevice(cur);
  }
  {
    std::lock_guard<std::mutex> lk(g->mu);
    g->abort.store(false); g->arrived = 0; g->left = 0;
    for (auto& rk : g->ranks) rk.calls = 0;
    std::fill(g->counts.begin(), g->counts.end(), (size_t)0);
    std::fill(g->index.begin(), g->index.end(), -1LL);
    std::fill(g->pub.begin(), g->pub.end(), nullptr);
  }
  std::vector<int> rcs((size_t)N, 0);
  std::vector<std::string> msgs((size_t)N);
  // a rank's call, with everything that can go wrong turned into a code + message and the group told about it
  auto body = [&](int r) {
    try {
      rcs[(size_t)r] = fn(g->child[(size_t)r], r);
      if (rcs[(size_t)r]) msgs[(size_t)r] = cuadmm_last_error();
    } catch (const std::bad_alloc&) {
      rcs[(size_t)r] = CUADMM_ERR_ALLOC; msgs[(size_t)r] = "out of host memory (std::bad_alloc)";
    } catch (const std::exception& e) {
      rcs[(size_t)r] = CUADMM_ERR_INTERNAL; msgs[(size_t)r] = std::string("exception: ") + e.what();
    } catch (...) {
      rcs[(size_t)r] = CUADMM_ERR_INTERNAL; msgs[(size_t)r] = "unknown exception";
    }
    if (rcs[(size_t)r]) g->raise_abort();
    g->rank_left();
  };
  std::vector<std::thread> th;
  try {
    for (int r = 1; r < N; ++r) th.emplace_back(body, r);
  } catch (...) {                                   // a thread could not be started: the ranks already running must not wait for it
    g->raise_abort();
    for (auto& t : th) t.join();
    set_error("duo group: could not start the host thread of rank %d", (int)th.size() + 1);
    return CUADMM_ERR_INTERNAL;
  }
  body(0);
  for (auto& t : th) t.join();
  // a rank that failed on its own account (not because the group was aborted under it) explains the failure best
  int first = -1;
  for (int r = 0; r < N; ++r) if (rcs[(size_t)r] && rcs[(size_t)r] != CUADMM_ERR_COMM) { first = r; break; }
  for (int r = 0; r < N && first < 0; ++r) if (rcs[(size_t)r]) first = r;
  if (first >= 0) {
    set_error("duo group, rank %d: %s", first, msgs[(size_t)first].c_str());
    return rcs[(size_t)first];
  }
  return CUADMM_OK;
}

}  // namespace cuadmm
