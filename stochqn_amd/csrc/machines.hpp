// machines.hpp -- the single-device state machines as the single-process multi-device front-end
// (group.cpp) calls them: one call = one run_* of the reference ABI on ONE shard, on the calling
// thread's current device.  Same arguments and return codes as include/stochqn.h:381-383.
#pragma once
#include "runtime.hpp"

namespace sqn {

int local_run_oLBFGS(real_t step_size, real_t x[], real_t grad[], real_t** req, task_enum* task, workspace_oLBFGS* w,
                     info_enum* iter_info);
int local_run_SQN(real_t step_size, real_t x[], real_t grad[], real_t hess_vec[], real_t** req, real_t** req_vec,
                  task_enum* task, workspace_SQN* w, info_enum* iter_info);
int local_run_adaQN(real_t step_size, real_t x[], real_t f, real_t grad[], real_t** req, task_enum* task,
                    workspace_adaQN* w, info_enum* iter_info);
// A shard hands the machine its SLICE of the caller's host x / grad / hess_vec, so the whole host-caller path (pinned copies,
// x not uploaded again, passes in slices under the transfers) works per shard; with this switch on for the calling thread,
// *req / *req_vec of workspace arrays come back as DEVICE pointers (the group copies them into the caller's arrays itself).
void set_thread_dev_requests(bool on);

// ---- group.cpp: n sharded over P devices inside one process, behind the unchanged ABI ------------
// Does this workspace run sharded?  (option "devices" >= 2 and the arrays are not device pointers)
bool group_applies(const bfgs_mem* b, int n);
bool group_owns(const void* s_mem);                    // a library-owned sharded workspace (initialize_* in group mode)
int group_run_oLBFGS(real_t step_size, real_t x[], real_t grad[], real_t** req, task_enum* task, workspace_oLBFGS* w,
                     info_enum* iter_info);
int group_run_SQN(real_t step_size, real_t x[], real_t grad[], real_t hess_vec[], real_t** req, real_t** req_vec,
                  task_enum* task, workspace_SQN* w, info_enum* iter_info);
int group_run_adaQN(real_t step_size, real_t x[], real_t f, real_t grad[], real_t** req, task_enum* task,
                    workspace_adaQN* w, info_enum* iter_info);
// library-owned sharded workspaces; NULL when the group mode is off for this n (caller falls back to one device)
bool group_mode_for(int n);
workspace_oLBFGS* group_initialize_oLBFGS(int n, size_t mem_size, real_t hess_init, real_t y_reg, real_t min_curvature,
                                          int check_nan, int nthreads);
workspace_SQN* group_initialize_SQN(int n, size_t mem_size, size_t bfgs_upd_freq, real_t min_curvature, int use_grad_diff,
                                    real_t y_reg, int check_nan, int nthreads);
workspace_adaQN* group_initialize_adaQN(int n, size_t mem_size, size_t fisher_size, size_t bfgs_upd_freq, real_t max_incr,
                                        real_t min_curvature, real_t scal_reg, real_t rmsprop_weight, int use_grad_diff,
                                        real_t y_reg, int check_nan, int nthreads);
void group_dealloc(const void* s_mem);                 // frees the shards, the workers and the token block
// context-management entry points applied to a group (return false when `key` is not a group)
bool group_release(const void* key);
void group_release_all();
bool group_invalidate(const void* key);
bool group_export(const void* key, int* rc);
int group_layout(const void* key, int shard, int* device, size_t* offset, size_t* count);
int group_bind(const void* key, int shard, real_t* x, real_t* grad, real_t* hess_vec);
int group_request(const void* key, int shard, real_t** req, real_t** req_vec);
int group_foreach(const void* key, void (*fn)(void*, int, int, size_t, size_t), void* user);
int group_shards(const void* key);                     // 0 = not a group
int group_reducer_kind(const void* key);               // Reducer::Kind of its shards

}  // namespace sqn
