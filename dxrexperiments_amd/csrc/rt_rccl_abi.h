// rt_rccl_abi.h -- the few RCCL entry points rt_dist.hip calls, as it declares them for dlopen / dlsym (the library is opened at run
// time so that a single-GPU box without RCCL still loads libdxrexperiments_amd.so).  rt_rccl_abi_check.cpp -- built by `make`, linked into
// nothing -- holds these declarations against <rccl/rccl.h> at COMPILE time: a drift in ncclCommInitRank / ncclUniqueId / the enums would
// otherwise only show on the first multi-GPU run.
#pragma once

#include <hip/hip_runtime.h>
#include <stddef.h>

namespace rt_rccl {

struct nccl_id { char internal[128]; };                     // ncclUniqueId
typedef void *nccl_comm;                                    // ncclComm_t
typedef int (*fn_get_unique_id)(nccl_id *);                 // ncclGetUniqueId
typedef int (*fn_comm_init_rank)(nccl_comm *, int, nccl_id, int);       // ncclCommInitRank (the id BY VALUE)
typedef int (*fn_comm_destroy)(nccl_comm);                  // ncclCommDestroy
typedef int (*fn_all_reduce)(const void *, void *, size_t, int, int, nccl_comm, hipStream_t);       // ncclAllReduce
typedef int (*fn_all_gather)(const void *, void *, size_t, int, nccl_comm, hipStream_t);            // ncclAllGather
typedef const char *(*fn_error_string)(int);                // ncclGetErrorString
constexpr int NCCL_FLOAT = 7, NCCL_SUM = 0;                 // ncclFloat, ncclSum

}  // namespace rt_rccl
