// Shared host/device helpers for libmyfa_gfx950.so (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <tuple>
#include <utility>

#include "myfa.h"

namespace myfa {

// Records a message for myfa_last_error() and returns `code`.
int fail(int code, const char* fmt, ...) __attribute__((format(printf, 2, 3)));

// Kernel launches go through launch_kernel(): it keeps the status hipLaunchKernel returns for THIS
// launch.  (hipGetLastError() after a <<<>>> launch reads and clears the thread's sticky error
// slot, which may hold an error some earlier HIP call of the process left there; the library
// never touches that slot.)  check_launch() reports the first failed launch since the previous
// check_launch() on this thread, without synchronising.
extern thread_local hipError_t t_launch_err;
int check_launch(const char* what);

template <typename... KArgs, typename... Args>
inline void launch_kernel(void (*kern)(KArgs...), dim3 grid, dim3 block, unsigned lds_bytes, hipStream_t s, Args&&... args) {
    static_assert(sizeof...(KArgs) == sizeof...(Args), "kernel argument count");
    std::tuple<KArgs...> vals(static_cast<KArgs>(std::forward<Args>(args))...);
    void* ptrs[sizeof...(KArgs) ? sizeof...(KArgs) : 1];
    std::apply([&](auto&... v) { int i = 0; ((ptrs[i++] = (void*)&v), ...); }, vals);
    const hipError_t e = hipLaunchKernel(reinterpret_cast<const void*>(kern), grid, block, ptrs, lds_bytes, s);
    if (e != hipSuccess && t_launch_err == hipSuccess) t_launch_err = e;
}

constexpr int kWave = 64;  // CDNA wavefront width

__host__ __device__ constexpr int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

// ---- wave64 reductions on DPP / ds_swizzle (no LDS traffic, no shuffles) ----
// Steps: xor-1 and xor-2 inside each quad (quad_perm), mirror inside 8 and inside
// 16 lanes (row_half_mirror, row_mirror), swap the 16-lane rows inside each half
// (ds_swizzle bit mode, xor 0x10), then combine the two 32-lane halves through
// readlane.  The result is wave-uniform.
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
    return __builtin_bit_cast(
        float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}

__device__ __forceinline__ float swizzle_xor16(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, v), 0x401F));
}

__device__ __forceinline__ float read_lane(float v, int lane) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane));
}

struct MaxOp {
    __device__ __forceinline__ float operator()(float a, float b) const { return fmaxf(a, b); }
};
struct SumOp {
    __device__ __forceinline__ float operator()(float a, float b) const { return a + b; }
};

template <typename Op>
__device__ __forceinline__ float wave_reduce(float x, Op op) {
    x = op(x, dpp_mov<0xB1>(x));   // quad_perm [1,0,3,2]
    x = op(x, dpp_mov<0x4E>(x));   // quad_perm [2,3,0,1]
    x = op(x, dpp_mov<0x141>(x));  // row_half_mirror
    x = op(x, dpp_mov<0x140>(x));  // row_mirror
    x = op(x, swizzle_xor16(x));
    return op(read_lane(x, 0), read_lane(x, 32));
}

// All-reduce over the four lanes l, l ^ 16, l ^ 32, l ^ 48 (one lane of each 16-lane row), in
// registers: v_permlane32_swap(x, x) leaves {x.lo | x.lo} and {x.hi | x.hi}, whose combination
// every lane holds reduced over the two halves; v_permlane16_swap(c, c) leaves {r0 r0 r2 r2} and
// {r1 r1 r3 r3} (16-lane rows of c), whose combination is the total everywhere.
template <typename Op>
__device__ __forceinline__ float four_rows_allreduce(float x, Op op) {
    const unsigned u = __builtin_bit_cast(unsigned, x);
    const auto r1 = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    const unsigned a1 = r1[0], b1 = r1[1];     // (copied out first: see attn_mfma_common.h)
    const float c = op(__builtin_bit_cast(float, a1), __builtin_bit_cast(float, b1));
    const unsigned cu = __builtin_bit_cast(unsigned, c);
    const auto r2 = __builtin_amdgcn_permlane16_swap(cu, cu, false, false);
    const unsigned a2 = r2[0], b2 = r2[1];
    return op(__builtin_bit_cast(float, a2), __builtin_bit_cast(float, b2));
}

__device__ __forceinline__ float wave_max(float x) { return wave_reduce(x, MaxOp{}); }
__device__ __forceinline__ float wave_sum(float x) { return wave_reduce(x, SumOp{}); }

}  // namespace myfa
