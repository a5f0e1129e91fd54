// Row softmax for gfx950: one HBM read and one HBM write per element.
//
// Stands in for softmax_kernel_1..5 of the reference
// (csrc/my_flash_attn/softmax.cu:20-456), which sweep each row three times
// (max, exp+sum, divide: 20 B of traffic per element).  Here the row is loaded
// once into registers with 128-bit loads, reduced with wave64 DPP reductions (and
// one LDS exchange when a row spans several waves), and written once: 8 B per
// element, so the kernel is bounded by HBM bandwidth.
//
// Three kernels:
//   softmax_rows_reg<TPR, V, VW>  row held in registers; TPR threads per row, V
//                                 items of VW floats per thread (covers w <= 32768)
//   softmax_rows_stream<VW>       longer rows: online (max, sum) pass, then a
//                                 second pass that re-reads and writes (12 B/elt)
// VW = 4 needs w % 4 == 0 (then every row base is 16-byte aligned given a 16-byte
// aligned tensor base); otherwise VW = 1.
#include "myfa_common.h"

#include <cstdio>
#include <cstdlib>

namespace myfa {
namespace {

constexpr float kLog2e = 1.4426950408889634f;

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int VW>
__device__ __forceinline__ void vload(float (&dst)[VW], const float* p) {
    if constexpr (VW == 4) {
        const f32x4 t = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p));
        dst[0] = t[0]; dst[1] = t[1]; dst[2] = t[2]; dst[3] = t[3];
    } else {
        dst[0] = __builtin_nontemporal_load(p);
    }
}

template <int VW>
__device__ __forceinline__ void vstore(float* p, const float (&src)[VW]) {
    if constexpr (VW == 4) {
        f32x4 t;
        t[0] = src[0]; t[1] = src[1]; t[2] = src[2]; t[3] = src[3];
        __builtin_nontemporal_store(t, reinterpret_cast<f32x4*>(p));
    } else {
        __builtin_nontemporal_store(src[0], p);
    }
}

// Combines one value per wave across the NW waves that share a row.
// `slot` is this row's region of the LDS scratch (NW floats).
template <int NW, typename Op>
__device__ __forceinline__ float rowgroup_reduce(float wave_val, float* slot, int wave_in_row, Op op) {
    if constexpr (NW == 1) {
        return wave_val;
    } else {
        if ((threadIdx.x & 63) == 0) slot[wave_in_row] = wave_val;
        __syncthreads();
        float r = slot[0];
#pragma unroll
        for (int i = 1; i < NW; ++i) r = op(r, slot[i]);
        __syncthreads();  // slot is reused by the next reduction
        return r;
    }
}

template <int TPR, int V, int VW>
__global__ void __launch_bounds__((TPR < 256 ? 256 : TPR))
softmax_rows_reg(const float* __restrict__ in, float* __restrict__ out, int64_t h, int64_t w) {
    constexpr int BLOCK = TPR < 256 ? 256 : TPR;
    constexpr int RPB = BLOCK / TPR;   // rows per workgroup
    constexpr int NW = TPR / 64;       // waves per row
    __shared__ float scratch[RPB * NW];

    const int t = threadIdx.x % TPR;
    const int row_in_blk = threadIdx.x / TPR;
    const int64_t row = (int64_t)blockIdx.x * RPB + row_in_blk;
    // A whole wave (TPR == 64) or the whole workgroup (TPR >= 256) shares `row`,
    // so this exit is uniform over everything that meets at a barrier below.
    if (row >= h) return;

    const float* src = in + row * w;
    float* dst = out + row * w;
    float* slot = scratch + row_in_blk * NW;
    const int wave_in_row = t >> 6;

    float x[V][VW];
    float mx = -INFINITY;
#pragma unroll
    for (int v = 0; v < V; ++v) {
        const int64_t col = ((int64_t)v * TPR + t) * VW;
        if (col < w) {
            vload<VW>(x[v], src + col);
        } else {
#pragma unroll
            for (int c = 0; c < VW; ++c) x[v][c] = -INFINITY;
        }
#pragma unroll
        for (int c = 0; c < VW; ++c) mx = fmaxf(mx, x[v][c]);
    }
    mx = rowgroup_reduce<NW>(wave_max(mx), slot, wave_in_row, MaxOp{});

    float sum = 0.f;
#pragma unroll
    for (int v = 0; v < V; ++v)
#pragma unroll
        for (int c = 0; c < VW; ++c) {
            // exp(x - mx) = 2^((x - mx) * log2e): the difference first, it is exact for
            // the elements near the max, which carry the sum; padding holds -inf -> 0
            x[v][c] = __builtin_amdgcn_exp2f((x[v][c] - mx) * kLog2e);
            sum += x[v][c];
        }
    sum = rowgroup_reduce<NW>(wave_sum(sum), slot, wave_in_row, SumOp{});

    const float inv = 1.0f / sum;
#pragma unroll
    for (int v = 0; v < V; ++v) {
        const int64_t col = ((int64_t)v * TPR + t) * VW;
        if (col < w) {
#pragma unroll
            for (int c = 0; c < VW; ++c) x[v][c] *= inv;
            vstore<VW>(dst + col, x[v]);
        }
    }
}

// Rows too long to hold in registers: one 1024-thread workgroup per row.
template <int VW>
__global__ void __launch_bounds__(1024)
softmax_rows_stream(const float* __restrict__ in, float* __restrict__ out, int64_t h, int64_t w) {
    constexpr int T = 1024, NW = T / 64;
    __shared__ float scratch[NW];
    const int64_t row = blockIdx.x;
    if (row >= h) return;
    const float* src = in + row * w;
    float* dst = out + row * w;
    const int t = threadIdx.x, wave = t >> 6;

    // pass 1: per-thread online (max, sum of 2^((x - max) * log2e))
    float mx = -INFINITY, sum = 0.f;
    for (int64_t col = (int64_t)t * VW; col < w; col += (int64_t)T * VW) {
        float x[VW];
        if constexpr (VW == 4) {
            const f32x4 q = *reinterpret_cast<const f32x4*>(src + col);
            x[0] = q[0]; x[1] = q[1]; x[2] = q[2]; x[3] = q[3];
        } else {
            x[0] = src[col];
        }
        float cm = x[0];
#pragma unroll
        for (int c = 1; c < VW; ++c) cm = fmaxf(cm, x[c]);
        if (cm > mx) {
            sum *= __builtin_amdgcn_exp2f((mx - cm) * kLog2e);  // mx = -inf first time: sum is 0
            mx = cm;
        }
#pragma unroll
        for (int c = 0; c < VW; ++c) sum += __builtin_amdgcn_exp2f((x[c] - mx) * kLog2e);
    }
    const float row_mx = rowgroup_reduce<NW>(wave_max(mx), scratch, wave, MaxOp{});
    // threads that saw no element have mx = -inf, sum = 0: contribute 0
    const float part = (mx == -INFINITY) ? 0.f : sum * __builtin_amdgcn_exp2f((mx - row_mx) * kLog2e);
    const float row_sum = rowgroup_reduce<NW>(wave_sum(part), scratch, wave, SumOp{});
    const float inv = 1.0f / row_sum;

    // pass 2: the row was just read, so most of it is still in L2 / Infinity Cache
    for (int64_t col = (int64_t)t * VW; col < w; col += (int64_t)T * VW) {
        float x[VW];
        vload<VW>(x, src + col);
#pragma unroll
        for (int c = 0; c < VW; ++c)
            x[c] = __builtin_amdgcn_exp2f((x[c] - row_mx) * kLog2e) * inv;
        vstore<VW>(dst + col, x);
    }
}

template <int TPR, int V, int VW>
void launch_reg(const float* in, float* out, int64_t h, int64_t w, hipStream_t s) {
    constexpr int BLOCK = TPR < 256 ? 256 : TPR;
    constexpr int RPB = BLOCK / TPR;
    hipLaunchKernelGGL((softmax_rows_reg<TPR, V, VW>), dim3((unsigned)cdiv(h, RPB)), dim3(BLOCK), 0, s,
                       in, out, h, w);
}

template <int VW>
bool dispatch_reg(const float* in, float* out, int64_t h, int64_t w, hipStream_t s, int tpr, int v) {
#define MYFA_CASE(T, VV) \
    if (tpr == T && v == VV) { launch_reg<T, VV, VW>(in, out, h, w, s); return true; }
    MYFA_CASE(64, 1) MYFA_CASE(64, 2) MYFA_CASE(64, 4) MYFA_CASE(64, 8)
    MYFA_CASE(256, 1) MYFA_CASE(256, 2) MYFA_CASE(256, 4) MYFA_CASE(256, 8)
    MYFA_CASE(512, 2) MYFA_CASE(512, 4) MYFA_CASE(512, 8)
    MYFA_CASE(1024, 2) MYFA_CASE(1024, 4) MYFA_CASE(1024, 8)
#undef MYFA_CASE
    return false;
}

// Picks threads-per-row and items-per-thread for `items` = w / VW items per row.
// Small rows: one wave per row, four rows per workgroup (no barrier at all).
// Larger rows: enough threads that each holds at most 8 items (32 floats).
void choose_shape(int64_t items, int* tpr, int* v) {
    struct Opt { int tpr, v; };
    static const Opt table[] = {
        {64, 1}, {64, 2}, {64, 4}, {64, 8},      // up to 512 items (w <= 2048 at VW=4)
        {256, 4}, {256, 8},                      // up to 2048 items (w <= 8192)
        {512, 8},                                // up to 4096 items (w <= 16384)
        {1024, 8},                               // up to 8192 items (w <= 32768)
    };
    for (const Opt& o : table)
        if ((int64_t)o.tpr * o.v >= items) { *tpr = o.tpr; *v = o.v; return; }
    *tpr = 0; *v = 0;
}

}  // namespace

int softmax_f32(const float* in, float* out, int64_t h, int64_t w, hipStream_t s) {
    if (h < 0 || w < 0) return fail(MYFA_E_INVALID, "softmax: negative size h=%lld w=%lld", (long long)h, (long long)w);
    if (h == 0 || w == 0) return MYFA_OK;
    if (!in || !out) return fail(MYFA_E_INVALID, "softmax: null pointer");
    if (h > 0x7fffffffLL) return fail(MYFA_E_UNSUPPORTED, "softmax: h=%lld exceeds the grid limit", (long long)h);
    const bool vec4 = (w % 4 == 0) && (reinterpret_cast<uintptr_t>(in) % 16 == 0) &&
                      (reinterpret_cast<uintptr_t>(out) % 16 == 0);
    const int vw = vec4 ? 4 : 1;
    int tpr = 0, v = 0;
    choose_shape(w / vw, &tpr, &v);
    // MYFA_SOFTMAX_SHAPE="tpr,v" overrides the choice (benchmark sweeps only).
    if (const char* e = std::getenv("MYFA_SOFTMAX_SHAPE")) {
        int a = 0, b = 0;
        if (std::sscanf(e, "%d,%d", &a, &b) == 2 && (int64_t)a * b * vw >= w) { tpr = a; v = b; }
    }
    bool done = false;
    if (tpr) done = vec4 ? dispatch_reg<4>(in, out, h, w, s, tpr, v) : dispatch_reg<1>(in, out, h, w, s, tpr, v);
    if (!done) {
        if (vec4) hipLaunchKernelGGL(softmax_rows_stream<4>, dim3((unsigned)h), dim3(1024), 0, s, in, out, h, w);
        else      hipLaunchKernelGGL(softmax_rows_stream<1>, dim3((unsigned)h), dim3(1024), 0, s, in, out, h, w);
    }
    return check_launch("softmax_rows");
}

}  // namespace myfa
