// SDPDuoSolver's "N devices from ONE process" mode (reference src/duo_solver.cu:487-577: the moment matrices are spread over
// device_num_requested GPUs by 1 + N host threads and copied between them peer to peer, src/utils/check_gpus.cu:29-43).
//
// Here a GPU is a rank of the block-sharded engine.  cuadmm_duo_init(..., device_num_requested = N) on a
// handle whose world is 1 builds a GROUP: the caller's handle becomes rank 0, N - 1 child engines are created with the same
// options (the option log is replayed) on devices 0 .. N-1 -- or all on the caller's device with option "duo_share_device" = 1 --
// and every collective of the engine goes through an in-process all-reduce: each rank copies its buffer to a page-locked,
// portable staging buffer on its own stream, the ranks meet at a barrier, every rank adds the N buffers IN RANK ORDER on the
// host (bit-identical sums on every rank, as the replicated solve requires) and copies the sum back.  The exchanged vector is
// [A X | sums | A (S - C)], 2m+2 doubles (or 4 scalars when constraints are owned): at these sizes the exchange is latency, and
// two PCIe hops of a few hundred KB cost what a peer-to-peer ring would (xGMI is not needed for correctness; RCCL cannot run
// two ranks on one device, which the one-GPU test of this mode needs).
//
// init and solve of the ranks run on N host threads (the caller's thread is rank 0); the getters of the caller's handle gather
// the shards.  A rank that fails raises the group's abort flag, so that the others leave their barrier with an error.
#include <hip/hip_runtime.h>

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
struct DuoRank { DuoGroup* g = nullptr; int rank = 0; double* stage = nullptr; double* sum = nullptr; size_t cap = 0; };

struct DuoGroup {
  int world = 1;
  std::vector<cuadmm_solver*> child;       // [0] = the caller's handle (not owned), [r >= 1] owned
  std::vector<DuoRank> ranks;
  std::mutex mu;
  std::condition_variable cv;
  int arrived = 0;
  long long generation = 0;
  std::atomic<bool> abort{false};
  std::vector<size_t> counts;              // what each rank brought to the current collective (must agree)
  long long n_allreduce = 0;

  // returns false when the group was aborted (a rank failed) or the ranks disagree on the length
  bool barrier() {
    std::unique_lock<std::mutex> lk(mu);
    if (abort.load()) return false;
    const long long gen = generation;
    if (++arrived == world) { arrived = 0; ++generation; cv.notify_all(); return !abort.load(); }
    cv.wait(lk, [&] { return generation != gen || abort.load(); });
    return !abort.load();
  }
  void raise_abort() {
    { std::lock_guard<std::mutex> lk(mu); abort.store(true); }
    cv.notify_all();
  }
};

static int duo_allreduce_hook(void* user, double* buf, size_t count, void* hip_stream) {
  DuoRank* me = static_cast<DuoRank*>(user);
  DuoGroup* g = me->g;
  hipStream_t st = static_cast<hipStream_t>(hip_stream);
  if (count == 0) return 0;
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
  g->counts[(size_t)me->rank] = count;
  if (!g->barrier()) return 1;
  for (int r = 0; r < g->world; ++r)
    if (g->counts[(size_t)r] != count) { g->raise_abort(); return 2; }      // ranks issued different collectives
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

DuoGroup* duo_group_of(void* p) { return static_cast<DuoGroup*>(p); }

void duo_group_destroy(void* p) {
  DuoGroup* g = duo_group_of(p);
  if (!g) return;
  for (size_t r = 1; r < g->child.size(); ++r) cuadmm_destroy(g->child[r]);
  for (auto& rk : g->ranks) {
    if (rk.stage) (void)hipHostFree(rk.stage);
    if (rk.sum) (void)hipHostFree(rk.sum);
  }
  delete g;
}

int duo_group_world(void* p) { return p ? duo_group_of(p)->world : 1; }
cuadmm_solver* duo_group_rank(void* p, int r) { DuoGroup* g = duo_group_of(p); return (g && r >= 0 && r < g->world) ? g->child[(size_t)r] : nullptr; }
long long duo_group_allreduces(void* p) { return p ? duo_group_of(p)->n_allreduce : 0; }

// Builds the group around `parent` (rank 0): children with the parent's options, the hook on every rank.  The caller then runs
// `fn(rank_handle)` on every rank through duo_group_run (init, solve).
int duo_group_create(cuadmm_solver* parent, int world, int parent_device, bool share_device,
                     const std::vector<std::pair<std::string, double>>& option_log, void** out) {
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) { set_error("duo_init: no HIP device"); return CUADMM_ERR_NO_DEVICE; }
  if (!share_device && world > ndev) {
    set_error("duo_init: device_num_requested = %d but %d device(s) are visible (option duo_share_device = 1 runs every engine on device %d)", world, ndev, parent_device);
    return CUADMM_ERR_INVALID;
  }
  DuoGroup* g = new DuoGroup();
  g->world = world;
  g->child.assign((size_t)world, nullptr);
  g->ranks.assign((size_t)world, DuoRank{});
  g->counts.assign((size_t)world, 0);
  g->child[0] = parent;
  for (int r = 0; r < world; ++r) { g->ranks[(size_t)r].g = g; g->ranks[(size_t)r].rank = r; }
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
    // check_gpus.cu:29-43 walks devices 0 .. N-1; rank r takes device r (the parent keeps its own)
    const int dev = share_device ? parent_device : (r == parent_device ? 0 : r);
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
  { std::lock_guard<std::mutex> lk(g->mu); g->abort.store(false); g->arrived = 0; }
  std::vector<int> rcs((size_t)N, 0);
  std::vector<std::string> msgs((size_t)N);
  std::vector<std::thread> th;
  for (int r = 1; r < N; ++r)
    th.emplace_back([&, r] {
      rcs[(size_t)r] = fn(g->child[(size_t)r], r);
      if (rcs[(size_t)r]) { msgs[(size_t)r] = cuadmm_last_error(); g->raise_abort(); }
    });
  rcs[0] = fn(g->child[0], 0);
  if (rcs[0]) { msgs[0] = cuadmm_last_error(); g->raise_abort(); }
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
