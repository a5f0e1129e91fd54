// Pieces shared by the softmax kernels: 16-byte element packs and their loads / stores.
#ifndef MYFA_SOFTMAX_COMMON_H_
#define MYFA_SOFTMAX_COMMON_H_
#include "myfa_common.h"

namespace myfa {
namespace {

constexpr float kLog2e = 1.4426950408889634f;

// VW elements of E as one vector (16 bytes when VW > 1)
template <typename E, int VW> struct Pack { typedef E type __attribute__((ext_vector_type(VW))); };
template <typename E> struct Pack<E, 1> { typedef E type; };
template <typename E> constexpr int vec_width() { return 16 / (int)sizeof(E); }

template <typename E, int VW, bool NT = true>
__device__ __forceinline__ void vload(float (&dst)[VW], const E* p) {
    using P = typename Pack<E, VW>::type;
    if constexpr (VW == 1) {
        dst[0] = (float)__builtin_nontemporal_load(p);
    } else {
        P t;
        if constexpr (NT) t = __builtin_nontemporal_load(reinterpret_cast<const P*>(p));
        else t = *reinterpret_cast<const P*>(p);
#pragma unroll
        for (int c = 0; c < VW; ++c) dst[c] = (float)t[c];
    }
}

template <typename E, int VW, bool NT = true>
__device__ __forceinline__ void vstore(E* p, const float (&src)[VW]) {
    using P = typename Pack<E, VW>::type;
    if constexpr (VW == 1) {
        __builtin_nontemporal_store((E)src[0], p);
    } else {
        P t;
#pragma unroll
        for (int c = 0; c < VW; ++c) t[c] = (E)src[c];
        if constexpr (NT) __builtin_nontemporal_store(t, reinterpret_cast<P*>(p));
        else *reinterpret_cast<P*>(p) = t;
    }
}

}  // namespace
}  // namespace myfa
#endif  // MYFA_SOFTMAX_COMMON_H_
