// Pieces shared by the softmax kernels: 16-byte element packs and their loads / stores.
#ifndef MYFA_SOFTMAX_COMMON_H_
#define MYFA_SOFTMAX_COMMON_H_
#include "myfa_common.h"

namespace myfa {

// Fused prologue: the softmax is taken of  scale * x + mask  (and, with causal_rows, keys after
// the row's query hidden).  Kernels carry it behind a template flag PRO; with PRO = false the
// record is ignored and the code is the plain softmax.
struct Prologue {
    float scale = 1.f;            // multiplies every element first
    const float* mask = nullptr;  // additive float32 [mask_outer, len, inner]; slice o % mask_outer serves outer index o
    int64_t mask_outer = 1;
    int64_t causal_rows = 0;      // > 0 (inner == 1 only): outer index o is query o % causal_rows, keys j > query -> -inf
};

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

// Values that cross workgroups inside one launch (the (max, sum) pairs of the split kernels) are
// written and read with device-scope accesses: coherent across the per-XCD L2s without any
// workgroup writing back or invalidating a whole cache.
__device__ __forceinline__ void put(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float get(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// Arrival counters of different rows / column blocks sit 256 bytes apart: atomics on one cache line
// are served one after the other (1024 arrivals on 8 adjacent counters took 34 us).
constexpr int kCountStride = 64;

// x[0..VW) = elements j .. j+VW-1 along the softmax dimension of outer index o (inner == 1)
template <int VW>
__device__ __forceinline__ void prologue_row(float (&x)[VW], const Prologue& p, int64_t o, int64_t len, int64_t j) {
    if (p.mask != nullptr) {
        const float* mrow = p.mask + (o % p.mask_outer) * len + j;
#pragma unroll
        for (int c = 0; c < VW; ++c) x[c] = __builtin_fmaf(x[c], p.scale, mrow[c]);
    } else {
#pragma unroll
        for (int c = 0; c < VW; ++c) x[c] *= p.scale;
    }
    if (p.causal_rows > 0) {
        const int64_t qi = o % p.causal_rows;
#pragma unroll
        for (int c = 0; c < VW; ++c) if (j + c > qi) x[c] = -INFINITY;
    }
}

// x[0..VW) = inner indices i .. i+VW-1 of row j (along the softmax dimension) of outer index o
template <int VW>
__device__ __forceinline__ void prologue_col(float (&x)[VW], const Prologue& p, int64_t o, int64_t len, int64_t inner, int64_t j, int64_t i) {
    if (p.mask != nullptr) {
        const float* m = p.mask + ((o % p.mask_outer) * len + j) * inner + i;
#pragma unroll
        for (int c = 0; c < VW; ++c) x[c] = __builtin_fmaf(x[c], p.scale, m[c]);
    } else {
#pragma unroll
        for (int c = 0; c < VW; ++c) x[c] *= p.scale;
    }
}

}  // namespace
}  // namespace myfa
#endif  // MYFA_SOFTMAX_COMMON_H_
