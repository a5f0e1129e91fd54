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
//                                 items of VW floats per thread (covers w <= 65536)
//   softmax_rows_stream<VW>       longer rows: online (max, sum) pass, then a
//                                 second pass that re-reads and writes (12 B/elt)
// An item is 16 bytes: VW = 4 floats, or 8 halves / bfloat16s (the element type E is a
// template parameter; arithmetic is float32 whatever E is).  With w % VW == 0 every row base is
// 16-byte aligned given a 16-byte aligned tensor base.  Otherwise (SHIFT) a row starts `a`
// elements past a 16-byte boundary, a different `a` from row to row: its items are then the aligned
// 16-byte groups that cover it, item i holding columns i*VW - a .. i*VW - a + VW - 1; only the
// first and the last item of a row are partial, and those are read and written element by element
// (nothing outside the row is touched).  VW = 1 is left for tensors whose base is not 16-byte aligned.
#include <algorithm>
#include <type_traits>
#include "softmax_common.h"
#include "tuning.h"


namespace myfa {
namespace {

// One item of a row under SHIFT: columns col0 .. col0 + VW - 1, of which those outside 0 .. w-1 do
// not belong to the row (they read as -inf and are not written).
template <typename E, int VW, bool NT, bool PRO, typename I>
__device__ __forceinline__ void load_item(float (&x)[VW], const E* src, I col0, int64_t row, I w, const Prologue& pro) {
    if constexpr (!PRO) {
        // The aligned 16 bytes are read whole even where they reach into the neighbouring row or, for
        // the last item of the tensor, up to 12 bytes past its end: an aligned 16-byte group never
        // crosses a page, so the read cannot fault; what is not the row's is replaced by -inf.
        vload<E, VW, NT>(x, src + col0);
        if (col0 < 0 || col0 + VW > w) {
#pragma unroll
            for (int c = 0; c < VW; ++c)
                if (col0 + c < 0 || col0 + c >= w) x[c] = -INFINITY;
        }
    } else if (col0 >= 0 && col0 + VW <= w) {
        vload<E, VW, NT>(x, src + col0);
        prologue_row<VW>(x, pro, row, w, col0);
    } else {
#pragma unroll
        for (int c = 0; c < VW; ++c) {
            const I col = col0 + c;
            x[c] = -INFINITY;
            if (col >= 0 && col < w) {
                float e[1] = {(float)src[col]};
                prologue_row<1>(e, pro, row, w, col);
                x[c] = e[0];
            }
        }
    }
}
template <typename E, int VW, bool NT, typename I>
__device__ __forceinline__ void store_item(E* dst, I col0, I w, const float (&x)[VW]) {
    if (col0 >= 0 && col0 + VW <= w) {
        vstore<E, VW, NT>(dst + col0, x);
    } else {
#pragma unroll
        for (int c = 0; c < VW; ++c)
            if (col0 + c >= 0 && col0 + c < w) dst[col0 + c] = (E)x[c];
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

template <typename E, int TPR, int V, int VW, int NTM = 3, int MAP = 0, bool PRO = false, bool SHIFT = false>
__global__ void __launch_bounds__((TPR < 256 ? 256 : TPR))
softmax_rows_reg(const E* __restrict__ in, E* __restrict__ out, int64_t h, int64_t w, Prologue pro) {
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

    const E* src = in + row * w;
    E* dst = out + row * w;
    float* slot = scratch + row_in_blk * NW;
    const int wave_in_row = t >> 6;

    // item index of (v, thread): MAP 0 interleaves the threads of the row (item v*TPR + t);
    // MAP 1 gives each wave V consecutive 64-item pieces (a contiguous V KiB at VW = 4)
    // (SHIFT: 32-bit columns -- a row held in registers is at most 65536 elements -- which saves the
    // registers its per-item bounds need; the aligned builds keep the 64-bit arithmetic they were
    // measured with: at 16384 x 16384 the 32-bit form of the same kernel is 5 % slower)
    using I = std::conditional_t<SHIFT, int, int64_t>;
    auto item = [&](int v) -> I {
        if constexpr (MAP == 0) return (I)v * TPR + t;
        else return ((I)(t >> 6) * V + v) * 64 + (t & 63);
    };
    const I wi = (I)w;
    // SHIFT: the row starts `shift` elements past a 16-byte boundary; item i is columns i*VW - shift ..
    const int shift = SHIFT ? (int)((row * w) % VW) : 0;
    float x[V][VW];
    float mx = -INFINITY;
#pragma unroll
    for (int v = 0; v < V; ++v) {
        const I col = item(v) * VW - shift;
        if (col < wi) {
            if constexpr (SHIFT) {
                load_item<E, VW, (NTM & 1) != 0, PRO>(x[v], src, col, row, wi, pro);
            } else {
                vload<E, VW, (NTM & 1) != 0>(x[v], src + col);
                if constexpr (PRO) prologue_row<VW>(x[v], pro, row, w, col);
            }
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
        const I col = item(v) * VW - shift;
        if (col < wi) {
#pragma unroll
            for (int c = 0; c < VW; ++c) x[v][c] *= inv;
            if constexpr (SHIFT) store_item<E, VW, (NTM & 2) != 0>(dst, col, wi, x[v]);
            else vstore<E, VW, (NTM & 2) != 0>(dst + col, x[v]);
        }
    }
}

// Rows too long to hold in registers: 1024-thread workgroups.
//   PHASE 0: one workgroup per row does both passes (grid = h).
// With a workspace, few long rows (h far below the CU count) are also split into `nsplit` slices
// of `ips` items, one workgroup each, in two launches (grid = h * nsplit, a row's slices adjacent):
//   PHASE 1: (max, sum) of the slice -> part[row][slice][2]; the workgroup of a row that finishes
//            last (a counter per row, zero at launch) merges the row's pairs into fin[row][2].
//            Nobody waits for anybody.  The pairs cross workgroups inside the launch, so they are
//            written and read with device-scope accesses (past the per-XCD L2s).
//   PHASE 2: re-reads its slice and writes, with the row's (max, sum) from fin.
// stats: float fin[h][2], then part[h][nsplit][2].
template <typename E, int VW, bool PRO = false, bool SHIFT = false, int PHASE = 0>
__global__ void __launch_bounds__(1024)
softmax_rows_stream(const E* __restrict__ in, E* __restrict__ out, int64_t h, int64_t w, Prologue pro,
                    int nsplit, int64_t ips, unsigned* __restrict__ count, float* __restrict__ stats) {
    constexpr int T = 1024, NW = T / 64;
    __shared__ float scratch[NW];
    const int64_t row = PHASE == 0 ? blockIdx.x : blockIdx.x / nsplit;
    const int split = PHASE == 0 ? 0 : (int)(blockIdx.x % nsplit);
    if (row >= h) return;
    const E* src = in + row * w;
    E* dst = out + row * w;
    const int t = threadIdx.x, wave = t >> 6;
    // item i is columns i*VW - shift .. (SHIFT: the row starts `shift` elements past a 16-byte boundary)
    const int shift = SHIFT ? (int)((row * w) % VW) : 0;
    const int64_t col_begin = PHASE == 0 ? -(int64_t)shift : (int64_t)split * ips * VW - shift;
    const int64_t col_end = PHASE == 0 ? w : (col_begin + ips * VW < w ? col_begin + ips * VW : w);

    float row_mx, inv;
    if constexpr (PHASE != 2) {
        // pass 1: per-thread online (max, sum of 2^((x - max) * log2e))
        float mx = -INFINITY, sum = 0.f;
        for (int64_t col = col_begin + (int64_t)t * VW; col < col_end; col += (int64_t)T * VW) {
            float x[VW];
            if constexpr (SHIFT) {
                load_item<E, VW, false, PRO>(x, src, col, row, w, pro);
            } else {
                vload<E, VW, false>(x, src + col);
                if constexpr (PRO) prologue_row<VW>(x, pro, row, w, col);
            }
            float cm = x[0];
#pragma unroll
            for (int c = 1; c < VW; ++c) cm = fmaxf(cm, x[c]);
            if (cm > mx) {
                sum *= __builtin_amdgcn_exp2f((mx - cm) * kLog2e);  // mx = -inf first time: sum is 0
                mx = cm;
            }
            // while everything this thread has seen is -inf there is nothing to add: (-inf) - (-inf) is
            // NaN, and a NaN in `sum` would survive the later rescale by 0 (leading -inf elements of a
            // row are ordinary input: left padding, sliding-window masks)
            if (mx != -INFINITY) {
#pragma unroll
                for (int c = 0; c < VW; ++c) sum += __builtin_amdgcn_exp2f((x[c] - mx) * kLog2e);
            }
        }
        row_mx = rowgroup_reduce<NW>(wave_max(mx), scratch, wave, MaxOp{});
        // threads that saw no element have mx = -inf, sum = 0: contribute 0
        const float part = (mx == -INFINITY) ? 0.f : sum * __builtin_amdgcn_exp2f((mx - row_mx) * kLog2e);
        const float row_sum = rowgroup_reduce<NW>(wave_sum(part), scratch, wave, SumOp{});
        inv = 1.0f / row_sum;

        if constexpr (PHASE == 1) {
            float* fin = stats + row * 2;
            float* parts = stats + (h + row * nsplit) * 2;
            __shared__ unsigned s_last;
            if (t == 0) {
                put(parts + 2 * split, row_mx);
                put(parts + 2 * split + 1, row_sum);
            }
            // the pair must have arrived (device-scope stores are acknowledged at the coherence point)
            // before this workgroup counts itself in: a workgroup-scope barrier alone does not wait for it
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (t == 0)
                s_last = __hip_atomic_fetch_add(count + row * kCountStride, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)(nsplit - 1);
            __syncthreads();
            if (!s_last) return;   // uniform over the workgroup
            // at most 1024 slices: one pair per thread
            const float pm = t < nsplit ? get(parts + 2 * t) : -INFINITY;
            const float pl = t < nsplit ? get(parts + 2 * t + 1) : 0.f;
            const float M = rowgroup_reduce<NW>(wave_max(pm), scratch, wave, MaxOp{});
            const float pls = (pm == -INFINITY) ? 0.f : pl * __builtin_amdgcn_exp2f((pm - M) * kLog2e);
            const float S = rowgroup_reduce<NW>(wave_sum(pls), scratch, wave, SumOp{});
            if (t == 0) { fin[0] = M; fin[1] = S; }
            return;
        }
    } else {
        row_mx = stats[row * 2];
        inv = 1.0f / stats[row * 2 + 1];
    }

    // pass 2: the row was just read, so most of it is still in L2 / Infinity Cache
    for (int64_t col = col_begin + (int64_t)t * VW; col < col_end; col += (int64_t)T * VW) {
        float x[VW];
        if constexpr (SHIFT) {
            load_item<E, VW, true, PRO>(x, src, col, row, w, pro);
        } else {
            vload<E, VW>(x, src + col);
            if constexpr (PRO) prologue_row<VW>(x, pro, row, w, col);
        }
#pragma unroll
        for (int c = 0; c < VW; ++c)
            x[c] = __builtin_amdgcn_exp2f((x[c] - row_mx) * kLog2e) * inv;
        if constexpr (SHIFT) store_item<E, VW, true>(dst, col, w, x);
        else vstore<E, VW>(dst + col, x);
    }
}

// One launch configuration of the register-resident kernel.
//   map : 0 interleaved items, 1 wave-contiguous items
//   nt  : bit 0 non-temporal loads, bit 1 non-temporal stores
//   pad : unused dynamic LDS, only there to cap the workgroups per CU
struct Shape { int tpr, v, map, nt; unsigned pad; };

template <typename E, int TPR, int V, int VW, bool PRO, bool SHIFT = false>
void launch_reg(const E* in, E* out, int64_t h, int64_t w, hipStream_t s, const Shape& sh, const Prologue& pro) {
    constexpr int BLOCK = TPR < 256 ? 256 : TPR;
    constexpr int RPB = BLOCK / TPR;
    const dim3 g((unsigned)cdiv(h, RPB)), b(BLOCK);
    if constexpr (SHIFT) {
        // one load/store policy and item map for rows that are not 16-byte aligned (fewer builds)
        launch_kernel((softmax_rows_reg<E, TPR, V, VW, 3, 0, PRO, true>), g, b, sh.pad, s, in, out, h, w, pro);
    } else if constexpr (PRO) {
        // one load/store policy and item map for the fused-prologue flavour (fewer builds)
        launch_kernel((softmax_rows_reg<E, TPR, V, VW, 3, 0, true>), g, b, sh.pad, s, in, out, h, w, pro);
    } else if constexpr (VW > 1) {
        if (sh.map == 1 && sh.nt == 2) launch_kernel((softmax_rows_reg<E, TPR, V, VW, 2, 1, PRO>), g, b, sh.pad, s, in, out, h, w, pro);
        else if (sh.map == 1)          launch_kernel((softmax_rows_reg<E, TPR, V, VW, 3, 1, PRO>), g, b, sh.pad, s, in, out, h, w, pro);
        else if (sh.nt == 2)           launch_kernel((softmax_rows_reg<E, TPR, V, VW, 2, 0, PRO>), g, b, sh.pad, s, in, out, h, w, pro);
        else                           launch_kernel((softmax_rows_reg<E, TPR, V, VW, 3, 0, PRO>), g, b, sh.pad, s, in, out, h, w, pro);
    } else {
        launch_kernel((softmax_rows_reg<E, TPR, V, VW, 3, 0, PRO>), g, b, sh.pad, s, in, out, h, w, pro);
    }
}

template <typename E, int VW, bool PRO, bool SHIFT = false>
bool dispatch_reg(const E* in, E* out, int64_t h, int64_t w, hipStream_t s, const Shape& sh, const Prologue& pro) {
#define MYFA_CASE(T, VV) \
    if (sh.tpr == T && sh.v == VV) { launch_reg<E, T, VV, VW, PRO, SHIFT>(in, out, h, w, s, sh, pro); return true; }
    MYFA_CASE(64, 1) MYFA_CASE(64, 2) MYFA_CASE(64, 4) MYFA_CASE(64, 8)
    MYFA_CASE(256, 1) MYFA_CASE(256, 2) MYFA_CASE(256, 4) MYFA_CASE(256, 8)
    MYFA_CASE(512, 2) MYFA_CASE(512, 4) MYFA_CASE(512, 8)
    MYFA_CASE(1024, 2) MYFA_CASE(1024, 4) MYFA_CASE(1024, 8)
    if constexpr (VW == 4) { MYFA_CASE(1024, 16) }
#undef MYFA_CASE
    return false;
}

// Picks the launch configuration for `items` = w / VW items per row of an h-row matrix.
// Small rows: one wave per row, four rows per workgroup (no barrier at all).  Larger rows:
// choices measured on MI355X at n x n, n = 4096 ... 32768 (tools/sweep_softmax*.py): many
// threads with few items each, wave-contiguous items, and -- for the two largest -- fewer
// resident waves than the register count would allow (at w = 16384 half of a 1024-thread
// workgroup's waves hold only padding; at 32768 unused LDS caps the workgroups per CU).
// Tensors that fit the Infinity Cache are read with plain loads, larger ones streamed
// with non-temporal loads; stores are always non-temporal.
// 16-bit elements put 8 values (8 registers) in an item, so they take the shapes with half the
// items per thread wherever the float32 choice would exceed 64 registers of row data.
Shape choose_shape(int64_t items, int64_t tensor_bytes, int vw, bool shifted = false) {
    Shape sh{0, 0, 0, tensor_bytes <= (128LL << 20) ? 2 : 3, 0};
    if (shifted) {
        // rows off the 16-byte grid come in every width: the smallest shape that holds the row
        // (the measured exceptions below belong to the power-of-two widths they were measured at)
        if (items <= 64)        { sh.tpr = 64;   sh.v = 1; }
        else if (items <= 128)  { sh.tpr = 64;   sh.v = 2; }
        else if (items <= 256)  { sh.tpr = 64;   sh.v = 4; }
        else if (items <= 512)  { if (vw == 8) { sh.tpr = 256; sh.v = 2; } else { sh.tpr = 64; sh.v = 8; } }
        else if (items <= 1024) { sh.tpr = 512;  sh.v = 2; }
        else if (items <= 2048) { sh.tpr = 1024; sh.v = 2; }
        else if (items <= 4096) { sh.tpr = 1024; sh.v = 4; }
        else if (items <= 8192) { sh.tpr = 1024; sh.v = 8; }
        else if (items <= 16384 && vw == 4) { sh.tpr = 1024; sh.v = 16; }
        return sh;
    }
    if (vw == 8) {
        if (items <= 64)        { sh.tpr = 64;   sh.v = 1; }
        else if (items <= 128)  { sh.tpr = 64;   sh.v = 2; }
        else if (items <= 256)  { sh.tpr = 64;   sh.v = 4; }
        else if (items <= 512)  { sh.tpr = 256;  sh.v = 2; }
        else if (items <= 1024) { sh.tpr = 512;  sh.v = 2; sh.map = 1; }
        else if (items <= 2048) { sh.tpr = 1024; sh.v = 2; sh.map = 1; }
        else if (items <= 4096) { sh.tpr = 1024; sh.v = 4; sh.map = 1; }
        else if (items <= 8192) { sh.tpr = 1024; sh.v = 8; sh.pad = 27000; }
        return sh;
    }
    if (items <= 64)        { sh.tpr = 64;   sh.v = 1; }
    else if (items <= 128)  { sh.tpr = 64;   sh.v = 2; }
    else if (items <= 256)  { sh.tpr = 64;   sh.v = 4; }
    else if (items <= 512)  { sh.tpr = 64;   sh.v = 8; }
    else if (items <= 1024) { sh.tpr = 512;  sh.v = 2; sh.map = 1; }
    else if (items <= 2048) { sh.tpr = 1024; sh.v = 2; sh.map = 1; }
    else if (items <= 4096) { sh.tpr = 1024; sh.v = 8; sh.map = 1; }
    else if (items <= 8192) { sh.tpr = 1024; sh.v = 8; sh.pad = 27000; }
    else if (items <= 16384) { sh.tpr = 1024; sh.v = 16; }
    return sh;
}

// Few long rows with a workspace: slices of a row over workgroups.  nsplit = 0: no split.
struct RowSplit {
    int nsplit = 0;
    int64_t ips = 0;                       // items per slice, a multiple of the 1024 threads
    size_t count_bytes = 0, bytes = 0;     // workspace: u32 count[h], 256 bytes apart (zero at launch); float fin[h][2], part[h][nsplit][2]
};
// items: 16-byte items per row (covering groups for rows off the 16-byte grid)
RowSplit rows_split_plan(int64_t h, int64_t items, int vw) {
    RowSplit p;
    if (items <= (vw == 4 ? 16384 : 8192) || h >= 256) return p;   // the register-resident kernel, or rows enough for the chip
    const int64_t nsplit = std::min<int64_t>({1024, cdiv((int64_t)1024, h), items / 2048});
    if (nsplit < 2) return p;
    p.ips = cdiv(cdiv(items, nsplit), (int64_t)1024) * 1024;
    p.nsplit = (int)cdiv(items, p.ips);
    p.count_bytes = (size_t)h * kCountStride * 4;
    p.bytes = p.count_bytes + (size_t)h * (size_t)(1 + p.nsplit) * 2 * sizeof(float);
    return p;
}

template <typename E, bool PRO>
int softmax_any(const E* in, E* out, int64_t h, int64_t w, const Prologue& pro, hipStream_t s,
                void* ws = nullptr, size_t ws_bytes = 0) {
    if (h < 0 || w < 0) return fail(MYFA_E_INVALID, "softmax: negative size h=%lld w=%lld", (long long)h, (long long)w);
    if (h == 0 || w == 0) return MYFA_OK;
    if (!in || !out) return fail(MYFA_E_INVALID, "softmax: null pointer");
    if (h > 0x7fffffffLL) return fail(MYFA_E_UNSUPPORTED, "softmax: h=%lld exceeds the grid limit", (long long)h);
    constexpr int VWV = vec_width<E>();
    const bool aligned = (reinterpret_cast<uintptr_t>(in) % 16 == 0) && (reinterpret_cast<uintptr_t>(out) % 16 == 0);
    const bool vec = aligned && (w % VWV == 0);
    // rows that start off a 16-byte boundary: the aligned groups that cover a row, up to one more than w / VW
    const bool shifted = aligned && !vec && w >= 2 * VWV;
    const int vw = (vec || shifted) ? VWV : 1;
    Shape sh = choose_shape(shifted ? cdiv(w + VWV - 1, (int64_t)VWV) : w / vw, h * w * (int64_t)sizeof(E), vw, shifted);
    // the tuning flavour of the library can force a launch shape (benchmark sweeps only)
    if (const int* f = tuning().softmax_shape; f[0] > 0 && (int64_t)f[0] * f[1] * vw >= w)
        sh = Shape{f[0], f[1], f[2], f[3], (unsigned)f[4]};
    bool done = false;
    if (sh.tpr) done = vec ? dispatch_reg<E, VWV, PRO>(in, out, h, w, s, sh, pro)
                   : shifted ? dispatch_reg<E, VWV, PRO, true>(in, out, h, w, s, sh, pro)
                             : dispatch_reg<E, 1, PRO>(in, out, h, w, s, sh, pro);
    if (!done) {
        const RowSplit p = (vec || shifted) ? rows_split_plan(h, shifted ? cdiv(w + VWV - 1, (int64_t)VWV) : w / VWV, VWV) : RowSplit{};
        if (ws != nullptr && p.nsplit > 1 && ws_bytes >= p.bytes) {
            if (const hipError_t e = hipMemsetAsync(ws, 0, p.count_bytes, s); e != hipSuccess)
                return fail(MYFA_E_LAUNCH, "softmax: clearing the workspace: %s", hipGetErrorString(e));
            unsigned* count = static_cast<unsigned*>(ws);
            float* stats = reinterpret_cast<float*>(static_cast<char*>(ws) + p.count_bytes);
            const dim3 g((unsigned)(h * p.nsplit)), b(1024);
            if (vec) {
                launch_kernel((softmax_rows_stream<E, VWV, PRO, false, 1>), g, b, 0, s, in, out, h, w, pro, p.nsplit, p.ips, count, stats);
                launch_kernel((softmax_rows_stream<E, VWV, PRO, false, 2>), g, b, 0, s, in, out, h, w, pro, p.nsplit, p.ips, count, stats);
            } else {
                launch_kernel((softmax_rows_stream<E, VWV, PRO, true, 1>), g, b, 0, s, in, out, h, w, pro, p.nsplit, p.ips, count, stats);
                launch_kernel((softmax_rows_stream<E, VWV, PRO, true, 2>), g, b, 0, s, in, out, h, w, pro, p.nsplit, p.ips, count, stats);
            }
        } else if (vec)   launch_kernel((softmax_rows_stream<E, VWV, PRO>), dim3((unsigned)h), dim3(1024), 0, s, in, out, h, w, pro, 1, (int64_t)0, (unsigned*)nullptr, (float*)nullptr);
        else if (shifted) launch_kernel((softmax_rows_stream<E, VWV, PRO, true>), dim3((unsigned)h), dim3(1024), 0, s, in, out, h, w, pro, 1, (int64_t)0, (unsigned*)nullptr, (float*)nullptr);
        else              launch_kernel((softmax_rows_stream<E, 1, PRO>), dim3((unsigned)h), dim3(1024), 0, s, in, out, h, w, pro, 1, (int64_t)0, (unsigned*)nullptr, (float*)nullptr);
    }
    return check_launch("softmax_rows");
}

}  // namespace

int softmax_f32(const float* in, float* out, int64_t h, int64_t w, hipStream_t s) {
    return softmax_any<float, false>(in, out, h, w, Prologue{}, s);
}

// dtype: MYFA_F32, MYFA_F16 or MYFA_BF16 (element type of both tensors; arithmetic is float32)
int softmax_rows(const void* in, void* out, int64_t h, int64_t w, int dtype, hipStream_t s) {
    switch (dtype) {
        case MYFA_F32:  return softmax_any<float, false>(static_cast<const float*>(in), static_cast<float*>(out), h, w, Prologue{}, s);
        case MYFA_F16:  return softmax_any<_Float16, false>(static_cast<const _Float16*>(in), static_cast<_Float16*>(out), h, w, Prologue{}, s);
        case MYFA_BF16: return softmax_any<__bf16, false>(static_cast<const __bf16*>(in), static_cast<__bf16*>(out), h, w, Prologue{}, s);
        default:        return fail(MYFA_E_INVALID, "softmax: unknown dtype tag %d", dtype);
    }
}

// The same with the fused prologue (scale, additive mask, causal rows); pro.mask / pro.causal_rows
// are indexed by the row number, so h is the `outer` of the [outer, len] view.
// ws: null, or at least softmax_rows_ws_bytes() bytes, 16-byte aligned (a smaller one is ignored)
int softmax_rows_pro(const void* in, void* out, int64_t h, int64_t w, int dtype, const Prologue& pro, hipStream_t s,
                     void* ws, size_t ws_bytes) {
    switch (dtype) {
        case MYFA_F32:  return softmax_any<float, true>(static_cast<const float*>(in), static_cast<float*>(out), h, w, pro, s, ws, ws_bytes);
        case MYFA_F16:  return softmax_any<_Float16, true>(static_cast<const _Float16*>(in), static_cast<_Float16*>(out), h, w, pro, s, ws, ws_bytes);
        case MYFA_BF16: return softmax_any<__bf16, true>(static_cast<const __bf16*>(in), static_cast<__bf16*>(out), h, w, pro, s, ws, ws_bytes);
        default:        return fail(MYFA_E_INVALID, "softmax: unknown dtype tag %d", dtype);
    }
}

// Workspace with which few long rows are split over workgroups; 0: they are not (short rows, rows
// enough for the chip).  Sized for 16-byte aligned tensors (others run element-wise, unsplit).
size_t softmax_rows_ws_bytes(int64_t h, int64_t w, int dtype) {
    if (h <= 0 || w <= 0) return 0;
    const int64_t vw = dtype == MYFA_F32 ? 4 : (dtype == MYFA_F16 || dtype == MYFA_BF16) ? 8 : 0;
    if (vw == 0) return 0;
    return rows_split_plan(h, w % vw == 0 ? w / vw : cdiv(w + vw - 1, vw), (int)vw).bytes;
}

}  // namespace myfa
