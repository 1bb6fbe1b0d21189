// librccl bound at run time (dlopen) by zip_mctx: libzip_hip.so has no link-time RCCL dependency, and a process that
// already carries an RCCL (PyTorch's) gets that copy.  The function-pointer types come from RCCL's own header where
// the toolchain has it (decltype of the real prototypes: a signature cannot drift), and are declared by hand
// otherwise; tests/test_rccl_prototypes.py compiles both forms side by side and static_asserts that every hand-written
// type is ABI-equivalent to the real one (same arity; identical parameter types, or an opaque pointer for a handle, or
// `int` for an enum whose underlying type is int) and that kUint8 == ncclUint8.
#pragma once
#include <hip/hip_runtime_api.h>
#include <stddef.h>

#if defined(__has_include)
#if __has_include(<rccl/rccl.h>) && !defined(ZIP_RCCL_HAND_DECLARED_ONLY)
#include <rccl/rccl.h>
#define ZIP_HAVE_RCCL_HEADER 1
#endif
#endif

namespace rccl_hand {  // what the library assumes of librccl when it is built without rccl.h
typedef int (*comm_init_all_t)(void **comms, int ndev, const int *devlist);
typedef int (*comm_destroy_t)(void *comm);
typedef int (*group_t)(void);
typedef int (*all_gather_t)(const void *send, void *recv, size_t count, int dtype, void *comm, hipStream_t st);
typedef int (*broadcast_t)(const void *send, void *recv, size_t count, int dtype, int root, void *comm, hipStream_t st);
typedef const char *(*err_str_t)(int);
typedef void *comm_t;
typedef int dtype_t;
constexpr int kUint8 = 1;  // ncclUint8 (= ncclChar + 1; rccl.h ncclDataType_t)
}  // namespace rccl_hand

namespace rccl {
#ifdef ZIP_HAVE_RCCL_HEADER
typedef decltype(&ncclCommInitAll) comm_init_all_t;
typedef decltype(&ncclCommDestroy) comm_destroy_t;
typedef decltype(&ncclGroupStart) group_t;
typedef decltype(&ncclAllGather) all_gather_t;
typedef decltype(&ncclBroadcast) broadcast_t;
typedef decltype(&ncclGetErrorString) err_str_t;
typedef ncclComm_t comm_t;
typedef ncclDataType_t dtype_t;
typedef ncclResult_t result_t;
constexpr ncclDataType_t kUint8 = ncclUint8;
#else
using namespace rccl_hand;
typedef int result_t;
#endif
}  // namespace rccl
