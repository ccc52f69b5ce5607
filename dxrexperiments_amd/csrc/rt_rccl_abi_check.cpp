// rt_rccl_abi_check.cpp -- compile-time check of rt_rccl_abi.h against the installed <rccl/rccl.h>.  Built by `make` (build/rt_rccl_abi_check.o),
// never linked into the product: it defines nothing.  Enumerations are passed as int and ncclComm_t as a pointer in the C ABI; everything else
// must match token for token.
#include <rccl/rccl.h>

#include <type_traits>

#include "rt_rccl_abi.h"

namespace {

using namespace rt_rccl;

template <class T> struct abi { using type = T; };
template <> struct abi<ncclResult_t> { using type = int; };
template <> struct abi<ncclDataType_t> { using type = int; };
template <> struct abi<ncclRedOp_t> { using type = int; };
template <> struct abi<ncclComm_t> { using type = nccl_comm; };
template <> struct abi<ncclComm_t *> { using type = nccl_comm *; };
template <> struct abi<ncclUniqueId> { using type = nccl_id; };
template <> struct abi<ncclUniqueId *> { using type = nccl_id *; };
template <class R, class... A> struct abi<R (*)(A...)> { using type = typename abi<R>::type (*)(typename abi<A>::type...); };
template <class F, class Mine> constexpr bool same_abi = std::is_same<typename abi<F>::type, Mine>::value;

static_assert(sizeof(ncclResult_t) == sizeof(int) && sizeof(ncclDataType_t) == sizeof(int) && sizeof(ncclRedOp_t) == sizeof(int), "RCCL enumerations are not int-sized");
static_assert(std::is_pointer<ncclComm_t>::value && sizeof(ncclComm_t) == sizeof(nccl_comm), "ncclComm_t is not a plain pointer");
static_assert(sizeof(ncclUniqueId) == sizeof(nccl_id) && alignof(ncclUniqueId) == alignof(nccl_id) && NCCL_UNIQUE_ID_BYTES == 128 &&
              std::is_trivially_copyable<ncclUniqueId>::value && std::is_standard_layout<ncclUniqueId>::value, "ncclUniqueId is not 128 plain bytes");
static_assert((int)ncclFloat == NCCL_FLOAT && (int)ncclSum == NCCL_SUM && (int)ncclSuccess == 0, "RCCL enumeration values moved");
static_assert(same_abi<decltype(&ncclGetUniqueId), fn_get_unique_id>, "ncclGetUniqueId");
static_assert(same_abi<decltype(&ncclCommInitRank), fn_comm_init_rank>, "ncclCommInitRank");
static_assert(same_abi<decltype(&ncclCommDestroy), fn_comm_destroy>, "ncclCommDestroy");
static_assert(same_abi<decltype(&ncclAllReduce), fn_all_reduce>, "ncclAllReduce");
static_assert(same_abi<decltype(&ncclAllGather), fn_all_gather>, "ncclAllGather");
static_assert(same_abi<decltype(&ncclGetErrorString), fn_error_string>, "ncclGetErrorString");

}  // namespace
