// SDPDuoSolver's N-devices-from-one-process mode (duo_group.hip): the caller's handle is rank 0 of a group of engines that run on
// host threads and exchange through an in-process all-reduce.
#pragma once
#include <functional>
#include <string>
#include <utility>
#include <vector>

struct cuadmm_solver;

namespace cuadmm {

int duo_group_create(cuadmm_solver* parent, int world, int parent_device, bool share_device,
                     const std::vector<std::pair<std::string, double>>& option_log, void** out);
void duo_group_destroy(void* group);
int duo_group_world(void* group);
cuadmm_solver* duo_group_rank(void* group, int r);
long long duo_group_allreduces(void* group);
// fn(rank handle, rank) on every rank (ranks >= 1 on their own host threads); the first failure's code and message
int duo_group_run(void* group, const std::function<int(cuadmm_solver*, int)>& fn);

}  // namespace cuadmm
