// SDPDuoSolver's N-devices-from-one-process mode (duo_group.hip): the caller's handle is rank 0 of a group of engines that run on
// host threads and exchange through an in-process all-reduce.
#pragma once
#include <functional>
#include <string>
#include <utility>
#include <vector>

struct cuadmm_solver;

namespace cuadmm {

// exchange: -1 = device-side exchange when every rank can read every other rank's device memory, else host-staged; 0 = host; 1 = device
int duo_group_create(cuadmm_solver* parent, int world, int parent_device, bool share_device, int exchange,
                     const std::vector<std::pair<std::string, double>>& option_log, void** out);
void duo_group_destroy(void* group);
int duo_group_world(void* group);
cuadmm_solver* duo_group_rank(void* group, int r);
long long duo_group_allreduces(void* group);
int duo_group_exchange(void* group);             // 1: device-side (peer reads), 0: host-staged
int duo_group_distinct_devices(void* group);
void duo_group_inject(void* group, long long v); // test hook, see DuoGroup::inject
// fn(rank handle, rank) on every rank (ranks >= 1 on their own host threads); the first failure's code and message
int duo_group_run(void* group, const std::function<int(cuadmm_solver*, int)>& fn);

}  // namespace cuadmm
