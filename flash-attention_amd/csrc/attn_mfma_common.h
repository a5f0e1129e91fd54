// Pieces shared by the f16/bf16 matrix-core attention kernels (forward and backward):
// vector types, the MFMA wrapper per element type, the transposing LDS read, the
// lane-half exchanges and the LDS tile image.
#ifndef MYFA_ATTN_MFMA_COMMON_H_
#define MYFA_ATTN_MFMA_COMMON_H_
#include <utility>

#include "myfa_common.h"

namespace myfa {
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

// run():   the compiler's MFMA; it places the accumulator (vector or accumulator register file)
//          and keeps the MFMA's result-latency rules itself.
// acc_v(), first_v(): the same instruction written out with the accumulator pinned to VECTOR
//          registers (c += a b, and c = a b).  For kernels that run one wave per SIMD: there the
//          compiler plans every MFMA result into accumulator registers and then moves tiles that
//          vector instructions consume back and forth.  The compiler does not see an MFMA in
//          these, so the caller keeps the result-latency rule: no vector instruction may read or
//          write c until two further MFMAs have been issued behind the last one that wrote it
//          (8 passes = 32 cycles each; the rule asks for 11 issue slots of 4 cycles).
// acc_a(): the same with the accumulator pinned to ACCUMULATOR registers, for a tile that nothing but
//          MFMAs touches in the loop (the same rule holds for whatever reads it afterwards).
template <typename T> struct Mma;
template <> struct Mma<__bf16> {
    using vec8 = bf16x8;
    using vec4 = bf16x4;
    static __device__ __forceinline__ f32x16 run(s16x8 a, vec8 b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(vec8, a), b, c, 0, 0, 0);
    }
    static __device__ __forceinline__ void acc_v([[maybe_unused]] s16x8 a, [[maybe_unused]] vec8 b, [[maybe_unused]] f32x16& c) {
#if defined(__HIP_DEVICE_COMPILE__)
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
#endif
    }
    static __device__ __forceinline__ void first_v([[maybe_unused]] s16x8 a, [[maybe_unused]] vec8 b, [[maybe_unused]] f32x16& c) {
#if defined(__HIP_DEVICE_COMPILE__)
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(c) : "v"(a), "v"(b));
#endif
    }
    static __device__ __forceinline__ void acc_a([[maybe_unused]] s16x8 a, [[maybe_unused]] vec8 b, [[maybe_unused]] f32x16& c) {
#if defined(__HIP_DEVICE_COMPILE__)
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
#endif
    }
};
template <> struct Mma<_Float16> {
    using vec8 = f16x8;
    using vec4 = f16x4;
    static __device__ __forceinline__ f32x16 run(s16x8 a, vec8 b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(vec8, a), b, c, 0, 0, 0);
    }
    static __device__ __forceinline__ void acc_v([[maybe_unused]] s16x8 a, [[maybe_unused]] vec8 b, [[maybe_unused]] f32x16& c) {
#if defined(__HIP_DEVICE_COMPILE__)
        asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
#endif
    }
    static __device__ __forceinline__ void first_v([[maybe_unused]] s16x8 a, [[maybe_unused]] vec8 b, [[maybe_unused]] f32x16& c) {
#if defined(__HIP_DEVICE_COMPILE__)
        asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=&v"(c) : "v"(a), "v"(b));
#endif
    }
    static __device__ __forceinline__ void acc_a([[maybe_unused]] s16x8 a, [[maybe_unused]] vec8 b, [[maybe_unused]] f32x16& c) {
#if defined(__HIP_DEVICE_COMPILE__)
        asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
#endif
    }
};

// ---- v_mfma_f32_16x16x32: D[16x16] += A[16x32] B[32x16].  Lane l = (i = l & 15, g = l >> 4) holds
// A[row i][k = 8g..8g+7], B[k = 8g..8g+7][col i] and D[row 4g + r][col i], r = 0..3.
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <typename T> struct Mma16;
template <> struct Mma16<__bf16> {
    using vec8 = bf16x8;
    using vec4 = bf16x4;
    static __device__ __forceinline__ f32x4 run(s16x8 a, vec8 b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(vec8, a), b, c, 0, 0, 0);
    }
};
template <> struct Mma16<_Float16> {
    using vec8 = f16x8;
    using vec4 = f16x4;
    static __device__ __forceinline__ f32x4 run(s16x8 a, vec8 b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(vec8, a), b, c, 0, 0, 0);
    }
};

// x0, x1: per-lane partial results for the lane's two queries (block 0, block 1), a query's
// four partials sitting in lanes i, i+16, i+32, i+48.  Returns both totals in every lane.
//   1. v_permlane32_swap(x0, x1) -> {x0.lo | x1.lo}, {x0.hi | x1.hi}; their combination holds
//      x0 reduced over the two halves in lanes 0-31 and x1 likewise in lanes 32-63.
//   2. v_permlane16_swap(c, c) -> {r0 r0 r2 r2}, {r1 r1 r3 r3} (16-lane rows of c); their
//      combination is the total of x0 in rows 0,1 and the total of x1 in rows 2,3.
//   3. v_permlane32_swap(e, e) -> {e.lo | e.lo}, {e.hi | e.hi}: both totals everywhere.
struct Pair { float a, b; };
template <class Op>
__device__ __forceinline__ Pair four_row_reduce(float x0, float x1, Op op) {
    const auto r1 = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, x0), __builtin_bit_cast(unsigned, x1), false, false);
    const unsigned a1 = r1[0], b1 = r1[1];
    const float c = op(__builtin_bit_cast(float, a1), __builtin_bit_cast(float, b1));
    const unsigned cu = __builtin_bit_cast(unsigned, c);
    const auto r2 = __builtin_amdgcn_permlane16_swap(cu, cu, false, false);
    const unsigned a2 = r2[0], b2 = r2[1];
    const float e = op(__builtin_bit_cast(float, a2), __builtin_bit_cast(float, b2));
    const unsigned eu = __builtin_bit_cast(unsigned, e);
    const auto r3 = __builtin_amdgcn_permlane32_swap(eu, eu, false, false);
    const unsigned a3 = r3[0], b3 = r3[1];
    return {__builtin_bit_cast(float, a3), __builtin_bit_cast(float, b3)};
}

// Compile-time loop: f(IC<0>{}), f(IC<1>{}), ..., f(IC<N-1>{}).
template <int V> struct IC { static constexpr int value = V; };
template <int... I, class F>
__device__ __forceinline__ void static_for_impl(std::integer_sequence<int, I...>, F&& f) { (f(IC<I>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) { static_for_impl(std::make_integer_sequence<int, N>{}, f); }

typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
typedef __attribute__((address_space(3))) void lds_void;

__device__ __forceinline__ s16x4 lds_read_tr(const char* p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p));
}

// Both lane halves' values of x, in every lane: {x[lane & 31], x[(lane & 31) + 32]}.
// v_permlane32_swap exchanges the upper half of its first operand with the lower
// half of its second.  The two results are copied out of the builtin's vector
// before any bit_cast: __builtin_bit_cast applied directly to an element of that
// vector reads element 0 for every index (hipcc 7.2).
struct HalfPair { float lo, hi; };
__device__ __forceinline__ HalfPair both_halves(float x) {
    const unsigned u = __builtin_bit_cast(unsigned, x);
    const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    const unsigned a = r[0], b = r[1];
    return {__builtin_bit_cast(float, a), __builtin_bit_cast(float, b)};
}
__device__ __forceinline__ float half_exchange_max(float x) {
    const HalfPair p = both_halves(x);
    return fmaxf(p.lo, p.hi);
}
__device__ __forceinline__ float half_exchange_sum(float x) {
    const HalfPair p = both_halves(x);
    return p.lo + p.hi;
}

// Maxima of values that come straight out of MFMA accumulators, as single instructions.  fmaxf()
// on such a value first quiets a possible signalling NaN (v_max_f32 x, x, x) because the compiler
// cannot know the value is canonical; these do not, and a NaN input gives the other operand,
// as the hardware's v_max does.
__device__ __forceinline__ float vmax3(float a, float b, float c) {
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ float vmax2(float a, float b) {
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
struct RawMaxOp {
    __device__ __forceinline__ float operator()(float a, float b) const { return vmax2(a, b); }
};

// x = KEY > t ? ninf : x, KEY an inline constant (0..64): one compare into VCC, one select.
template <int KEY>
__device__ __forceinline__ float hide_past(float x, [[maybe_unused]] int t, [[maybe_unused]] float ninf) {
    static_assert(KEY >= 0 && KEY <= 64, "inline constant");
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("v_cmp_gt_i32 vcc, %2, %1\n\tv_cndmask_b32 %0, %0, %3, vcc" : "+v"(x) : "v"(t), "n"(KEY), "v"(ninf) : "vcc");
#endif
    return x;
}

// LDS image of a [rows][D] 16-bit tile made of 8-row x 64-byte subtiles (c = 16-byte chunk of
// the row).  Row reads (ds_read_b128: row = lane & 31, chunk 2*kstep + (lane >> 5)) and
// transposing reads (ds_read_b64_tr_b16) are both bank-conflict-free on it.
template <int D> __device__ __forceinline__ constexpr int img_off(int row, int ch) {
    return 16 * D * (row >> 3) + 512 * (ch >> 2) + 64 * (row & 7) + 16 * ((ch & 3) ^ ((row >> 2) & 3));
}

}  // namespace
}  // namespace myfa
#endif  // MYFA_ATTN_MFMA_COMMON_H_
