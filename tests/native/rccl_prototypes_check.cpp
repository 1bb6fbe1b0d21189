// Compile-only check (tests/test_rccl_prototypes.py): the hand-declared librccl prototypes that zinc_amd/csrc/rccl_dyn.h
// falls back to when RCCL's header is absent are ABI-equivalent to the real ones of <rccl/rccl.h>, and the data-type
// constant matches.  (With the header present the library uses decltype of the real prototypes anyway.)
#include <type_traits>

#include "rccl_dyn.h"

#ifndef ZIP_HAVE_RCCL_HEADER
#error "rccl/rccl.h not found: nothing to check against"
#endif

template <class A, class B, class = void>
struct abi_same : std::is_same<A, B> {};
// an opaque handle (or a pointer to one): object pointers share one representation; the pointee's constness must agree
template <class A, class B>
struct abi_same<A *, B *, std::enable_if_t<!std::is_same<A *, B *>::value>>
    : std::integral_constant<bool, std::is_const<A>::value == std::is_const<B>::value && !std::is_function<A>::value &&
                                       !std::is_function<B>::value> {};
// an enum passed (or returned) as an integer of its width
template <class A, class B>
struct abi_same<A, B, std::enable_if_t<std::is_enum<B>::value && !std::is_same<A, B>::value>>
    : std::integral_constant<bool, std::is_integral<A>::value && sizeof(A) == sizeof(B)> {};

template <class F, class G>
struct fn_same : std::false_type {};
template <class R1, class... A1, class R2, class... A2>
struct fn_same<R1 (*)(A1...), R2 (*)(A2...)> {
    static constexpr bool arity = sizeof...(A1) == sizeof...(A2);
    template <bool ok, class = void>
    struct args : std::false_type {};
    template <class Dummy>
    struct args<true, Dummy> : std::conjunction<abi_same<A1, A2>...> {};
    static constexpr bool value = arity && abi_same<R1, R2>::value && args<arity>::value;
};

static_assert(fn_same<rccl_hand::comm_init_all_t, decltype(&ncclCommInitAll)>::value, "ncclCommInitAll");
static_assert(fn_same<rccl_hand::comm_destroy_t, decltype(&ncclCommDestroy)>::value, "ncclCommDestroy");
static_assert(fn_same<rccl_hand::group_t, decltype(&ncclGroupStart)>::value, "ncclGroupStart");
static_assert(fn_same<rccl_hand::group_t, decltype(&ncclGroupEnd)>::value, "ncclGroupEnd");
static_assert(fn_same<rccl_hand::all_gather_t, decltype(&ncclAllGather)>::value, "ncclAllGather");
static_assert(fn_same<rccl_hand::broadcast_t, decltype(&ncclBroadcast)>::value, "ncclBroadcast");
static_assert(fn_same<rccl_hand::err_str_t, decltype(&ncclGetErrorString)>::value, "ncclGetErrorString");
static_assert(rccl_hand::kUint8 == (int)ncclUint8, "ncclUint8");
static_assert(sizeof(ncclDataType_t) == sizeof(int) && sizeof(ncclResult_t) == sizeof(int), "enum width");
static_assert(sizeof(ncclComm_t) == sizeof(void *), "handle width");
// ... and the ones the library really uses ARE the real ones
static_assert(std::is_same<rccl::all_gather_t, decltype(&ncclAllGather)>::value, "decltype form");
static_assert(std::is_same<rccl::broadcast_t, decltype(&ncclBroadcast)>::value, "decltype form");

int main() { return 0; }
