// Softmax along a middle dimension: the tensor is viewed as [outer][len][inner], dense, and
// out[o, :, i] = softmax(in[o, :, i]) for every (o, i) -- torch.softmax(x, dim) for any dim of a
// contiguous tensor (inner = 1 is the row softmax of softmax_rows.hip and is sent there).
//
// The reference's softmax is rows of a 2-D matrix only (csrc/api.cpp:110-166,
// csrc/my_flash_attn/softmax.cu:20-456); this is the generalisation of that entry point.
//
// A workgroup is TX x TY threads (256; 512 or 1024 for long streamed columns): TX threads side by side along `inner`, each owning one
// 16-byte item (VW elements) of a row, so a row segment of TX * 16 bytes is read at once; TY
// threads share the rows of the column block (row ty, ty + TY, ...).  Per column the TY partial
// (max, sum) pairs meet in LDS.
//   softmax_cols_reg<R>   len <= R * TY: the column block is held in registers, 1 read + 1 write
//   softmax_cols_stream   any len: online (max, sum) pass, then a pass that re-reads and writes
//                         (256 threads, or 512 / 1024 for long columns)
// VW = 4 floats or 8 halves needs inner % VW == 0 and 16-byte aligned bases; otherwise VW = 1.
#include "softmax_common.h"

namespace myfa {
int softmax_rows(const void* in, void* out, int64_t h, int64_t w, int dtype, hipStream_t s);
int softmax_rows_pro(const void* in, void* out, int64_t h, int64_t w, int dtype, const Prologue& pro, hipStream_t s);
namespace {

// Column statistics across the TY threads of a column: red is [TY][TX * VW] floats.
template <int TX, int TY, int VW, typename Op>
__device__ __forceinline__ void column_reduce(float (&v)[VW], float* red, int tx, int ty, Op op) {
    constexpr int W = TX * VW;
#pragma unroll
    for (int c = 0; c < VW; ++c) red[ty * W + c * TX + tx] = v[c];
    __syncthreads();
    if constexpr (TY <= 16) {
#pragma unroll
        for (int c = 0; c < VW; ++c) {
            float r = red[c * TX + tx];
#pragma unroll
            for (int t = 1; t < TY; ++t) r = op(r, red[t * W + c * TX + tx]);
            v[c] = r;
        }
    } else {
#pragma unroll
        for (int half = TY / 2; half >= 1; half >>= 1) {
            if (ty < half) {
#pragma unroll
                for (int c = 0; c < VW; ++c) red[ty * W + c * TX + tx] = op(red[ty * W + c * TX + tx], red[(ty + half) * W + c * TX + tx]);
            }
            __syncthreads();
        }
#pragma unroll
        for (int c = 0; c < VW; ++c) v[c] = red[c * TX + tx];
    }
    __syncthreads();   // red is reused by the next reduction
}

template <typename E, int TX, int VW, int R, bool PRO = false>
__global__ void __launch_bounds__(256)
softmax_cols_reg(const E* __restrict__ in, E* __restrict__ out, int64_t len, int64_t inner, int64_t nchunk, Prologue pro) {
    constexpr int TY = 256 / TX;
    __shared__ float red[256 * VW];
    const int tx = threadIdx.x % TX, ty = threadIdx.x / TX;
    const int64_t o = blockIdx.x / nchunk, chunk = blockIdx.x % nchunk;
    const int64_t col = (chunk * TX + tx) * VW;
    const bool live = col < inner;                       // (threads past the edge still meet the barriers)
    const E* src = in + o * len * inner + col;
    E* dst = out + o * len * inner + col;

    float x[R][VW], mx[VW];
#pragma unroll
    for (int c = 0; c < VW; ++c) mx[c] = -INFINITY;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int64_t row = ty + (int64_t)r * TY;
        if (live && row < len) {
            vload<E, VW>(x[r], src + row * inner);
            if constexpr (PRO) prologue_col<VW>(x[r], pro, o, len, inner, row, col);
        } else {
#pragma unroll
            for (int c = 0; c < VW; ++c) x[r][c] = -INFINITY;
        }
#pragma unroll
        for (int c = 0; c < VW; ++c) mx[c] = fmaxf(mx[c], x[r][c]);
    }
    column_reduce<TX, TY, VW>(mx, red, tx, ty, MaxOp{});

    float sum[VW];
#pragma unroll
    for (int c = 0; c < VW; ++c) sum[c] = 0.f;
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int c = 0; c < VW; ++c) {
            x[r][c] = __builtin_amdgcn_exp2f((x[r][c] - mx[c]) * kLog2e);    // padding holds -inf -> 0
            sum[c] += x[r][c];
        }
    column_reduce<TX, TY, VW>(sum, red, tx, ty, SumOp{});

#pragma unroll
    for (int c = 0; c < VW; ++c) sum[c] = 1.0f / sum[c];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int64_t row = ty + (int64_t)r * TY;
        if (live && row < len) {
#pragma unroll
            for (int c = 0; c < VW; ++c) x[r][c] *= sum[c];
            vstore<E, VW>(dst + row * inner, x[r]);
        }
    }
}

// Any len: BLOCK = 256, 512 or 1024 threads, rows taken U at a time per thread so that U loads are in
// flight and the running max is refreshed once per U rows.
template <typename E, int TX, int VW, int BLOCK, bool PRO = false>
__global__ void __launch_bounds__(BLOCK)
softmax_cols_stream(const E* __restrict__ in, E* __restrict__ out, int64_t len, int64_t inner, int64_t nchunk, Prologue pro) {
    constexpr int TY = BLOCK / TX, U = VW == 8 ? 2 : 4;
    __shared__ float red[BLOCK * VW];
    const int tx = threadIdx.x % TX, ty = threadIdx.x / TX;
    const int64_t o = blockIdx.x / nchunk, chunk = blockIdx.x % nchunk;
    const int64_t col = (chunk * TX + tx) * VW;
    const bool live = col < inner;
    const E* src = in + o * len * inner + col;
    E* dst = out + o * len * inner + col;

    // pass 1: per-thread online (max, sum of 2^((x - max) * log2e)) of each of its VW columns
    float mx[VW], sum[VW];
#pragma unroll
    for (int c = 0; c < VW; ++c) { mx[c] = -INFINITY; sum[c] = 0.f; }
    if (live) {
        for (int64_t row0 = ty; row0 < len; row0 += (int64_t)U * TY) {
            float x[U][VW];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int64_t row = row0 + (int64_t)u * TY;
                if (row < len) {
                    vload<E, VW, false>(x[u], src + row * inner);
                    if constexpr (PRO) prologue_col<VW>(x[u], pro, o, len, inner, row, col);
                } else {
#pragma unroll
                    for (int c = 0; c < VW; ++c) x[u][c] = -INFINITY;
                }
            }
#pragma unroll
            for (int c = 0; c < VW; ++c) {
                float cm = x[0][c];
#pragma unroll
                for (int u = 1; u < U; ++u) cm = fmaxf(cm, x[u][c]);
                if (cm > mx[c]) {
                    sum[c] *= __builtin_amdgcn_exp2f((mx[c] - cm) * kLog2e);   // mx = -inf first time: sum is 0
                    mx[c] = cm;
                }
                // rows past the end hold -inf and add 0.  While everything seen of a column is -inf
                // there is nothing to add: (-inf) - (-inf) is NaN and would survive the later rescale
                // by 0 (leading -inf rows are ordinary input: a causal mask along this dimension); only
                // a column that is -inf throughout comes out NaN, through 0 * inf in the second pass
                if (mx[c] != -INFINITY) {
#pragma unroll
                    for (int u = 0; u < U; ++u) sum[c] += __builtin_amdgcn_exp2f((x[u][c] - mx[c]) * kLog2e);
                }
            }
        }
    }
    float cmx[VW];
#pragma unroll
    for (int c = 0; c < VW; ++c) cmx[c] = mx[c];
    column_reduce<TX, TY, VW>(cmx, red, tx, ty, MaxOp{});
    // threads that saw no row have mx = -inf, sum = 0: they contribute 0
#pragma unroll
    for (int c = 0; c < VW; ++c)
        sum[c] = (mx[c] == -INFINITY) ? 0.f : sum[c] * __builtin_amdgcn_exp2f((mx[c] - cmx[c]) * kLog2e);
    column_reduce<TX, TY, VW>(sum, red, tx, ty, SumOp{});

    // pass 2: re-read (what is still in L2 / Infinity Cache comes from there) and write
    if (live) {
#pragma unroll
        for (int c = 0; c < VW; ++c) sum[c] = 1.0f / sum[c];
        for (int64_t row0 = ty; row0 < len; row0 += (int64_t)U * TY) {
            float x[U][VW];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int64_t row = row0 + (int64_t)u * TY;
                if (row < len) {
                    vload<E, VW>(x[u], src + row * inner);
                    if constexpr (PRO) prologue_col<VW>(x[u], pro, o, len, inner, row, col);
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int64_t row = row0 + (int64_t)u * TY;
                if (row < len) {
#pragma unroll
                    for (int c = 0; c < VW; ++c) x[u][c] = __builtin_amdgcn_exp2f((x[u][c] - cmx[c]) * kLog2e) * sum[c];
                    vstore<E, VW>(dst + row * inner, x[u]);
                }
            }
        }
    }
}

// Launch choice.  items = 16-byte items (or single elements) per row.
//  * Column block in registers (1 read + 1 write) whenever len fits RMAX rows per thread for some
//    row-segment width TX >= 8 items that still gives the chip enough workgroups -- the widest such
//    TX -- or the tensor is small anyway; 4-item segments (64 bytes, half an L2 line) only for rows
//    that narrow.
//  * Otherwise streamed: 1024 threads for long columns, the widest TX that leaves one workgroup
//    per CU; never narrower than 8 items (128 bytes, an L2 line) when the row has them.
template <typename E, int VW, bool PRO>
void launch_cols(const E* in, E* out, int64_t outer, int64_t len, int64_t inner, const Prologue& pro, hipStream_t s) {
    constexpr int RMAX = 64 / VW;      // at most 64 floats of column data per thread
    const int64_t items = cdiv(inner, (int64_t)VW);
    const bool small = outer * len * inner * (int64_t)sizeof(E) <= (8LL << 20);
    auto blocks = [&](int tx) { return outer * cdiv(items, (int64_t)tx); };
#define MYFA_REG(TXV) \
    if ((TXV == 4 ? (items <= 4 || small) : items > TXV / 2) && cdiv(len, (int64_t)(256 / TXV)) <= RMAX && (small || blocks(TXV) >= 512 || TXV == 4)) { \
        const int64_t nchunk = cdiv(items, (int64_t)TXV); \
        const dim3 g((unsigned)blocks(TXV)), b(256); \
        if (cdiv(len, (int64_t)(256 / TXV)) <= RMAX / 4) launch_kernel((softmax_cols_reg<E, TXV, VW, RMAX / 4, PRO>), g, b, 0, s, in, out, len, inner, nchunk, pro); \
        else launch_kernel((softmax_cols_reg<E, TXV, VW, RMAX, PRO>), g, b, 0, s, in, out, len, inner, nchunk, pro); \
        return; \
    }
    MYFA_REG(64) MYFA_REG(16) MYFA_REG(8) MYFA_REG(4)
#undef MYFA_REG
    const bool big = len >= 2048;
    constexpr int BIG = VW == 8 ? 512 : 1024;      // (eight-element items need the registers of a smaller workgroup)
#define MYFA_STREAM(TXV, LAST) \
    if (LAST || (items > TXV / 2 && blocks(TXV) >= 256)) { \
        const int64_t nchunk = cdiv(items, (int64_t)TXV); \
        const dim3 g((unsigned)blocks(TXV)); \
        if (big) launch_kernel((softmax_cols_stream<E, TXV, VW, BIG, PRO>), g, dim3(BIG), 0, s, in, out, len, inner, nchunk, pro); \
        else launch_kernel((softmax_cols_stream<E, TXV, VW, 256, PRO>), g, dim3(256), 0, s, in, out, len, inner, nchunk, pro); \
        return; \
    }
    MYFA_STREAM(64, false) MYFA_STREAM(16, false) MYFA_STREAM(8, items >= 8) MYFA_STREAM(4, true)
#undef MYFA_STREAM
}

template <typename E, bool PRO>
int softmax_cols_any(const E* in, E* out, int64_t outer, int64_t len, int64_t inner, const Prologue& pro, hipStream_t s) {
    constexpr int VWV = vec_width<E>();
    const bool vec = (inner % VWV == 0) && (reinterpret_cast<uintptr_t>(in) % 16 == 0) &&
                     (reinterpret_cast<uintptr_t>(out) % 16 == 0);
    if (outer * cdiv(inner, (int64_t)(vec ? VWV : 1) * 4) > 0x7fffffffLL)
        return fail(MYFA_E_UNSUPPORTED, "softmax: outer=%lld x inner=%lld exceeds the grid limit", (long long)outer, (long long)inner);
    if (vec) launch_cols<E, VWV, PRO>(in, out, outer, len, inner, pro, s);
    else launch_cols<E, 1, PRO>(in, out, outer, len, inner, pro, s);
    return check_launch("softmax_cols");
}

}  // namespace

namespace {
int check_sizes(const void* in, const void* out, int64_t outer, int64_t len, int64_t inner, int dtype, bool& empty) {
    empty = false;
    if (outer < 0 || len < 0 || inner < 0)
        return fail(MYFA_E_INVALID, "softmax: negative size outer=%lld len=%lld inner=%lld", (long long)outer, (long long)len, (long long)inner);
    if (dtype != MYFA_F32 && dtype != MYFA_F16 && dtype != MYFA_BF16) return fail(MYFA_E_INVALID, "softmax: unknown dtype tag %d", dtype);
    if (outer == 0 || len == 0 || inner == 0) { empty = true; return MYFA_OK; }
    if (!in || !out) return fail(MYFA_E_INVALID, "softmax: null pointer");
    return MYFA_OK;
}
}  // namespace

// dtype: MYFA_F32, MYFA_F16 or MYFA_BF16 (element type of both tensors; arithmetic is float32)
int softmax_strided(const void* in, void* out, int64_t outer, int64_t len, int64_t inner, int dtype, hipStream_t s) {
    bool empty;
    if (const int rc = check_sizes(in, out, outer, len, inner, dtype, empty); rc != MYFA_OK || empty) return rc;
    if (inner == 1) return softmax_rows(in, out, outer, len, dtype, s);
    const Prologue none{};
    switch (dtype) {
        case MYFA_F32:  return softmax_cols_any<float, false>(static_cast<const float*>(in), static_cast<float*>(out), outer, len, inner, none, s);
        case MYFA_F16:  return softmax_cols_any<_Float16, false>(static_cast<const _Float16*>(in), static_cast<_Float16*>(out), outer, len, inner, none, s);
        default:        return softmax_cols_any<__bf16, false>(static_cast<const __bf16*>(in), static_cast<__bf16*>(out), outer, len, inner, none, s);
    }
}

// softmax(scale * x + mask) along `len`, optionally with the causal mask of square logits.
//   mask: float32 [mask_outer, len, inner] or null; outer index o uses slice o % mask_outer
//   causal: inner must be 1 and outer a multiple of len: row o is query o % len of a [.., len, len] tensor
int softmax_strided_pro(const void* in, void* out, int64_t outer, int64_t len, int64_t inner, int dtype,
                        float scale, const float* mask, int64_t mask_outer, bool causal, hipStream_t s) {
    bool empty;
    if (const int rc = check_sizes(in, out, outer, len, inner, dtype, empty); rc != MYFA_OK) return rc;
    if (mask != nullptr && (mask_outer < 1 || (!empty && outer % mask_outer != 0)))
        return fail(MYFA_E_INVALID, "softmax: mask_outer=%lld must be >= 1 and divide outer=%lld", (long long)mask_outer, (long long)outer);
    if (causal && (inner != 1 || (len > 0 && outer % len != 0)))
        return fail(MYFA_E_UNSUPPORTED, "softmax: the causal flag needs the last dimension of square logits "
                    "(inner == 1, outer a multiple of len; got outer=%lld len=%lld inner=%lld)", (long long)outer, (long long)len, (long long)inner);
    if (empty) return MYFA_OK;
    Prologue pro;
    pro.scale = scale;
    pro.mask = mask;
    pro.mask_outer = mask != nullptr ? mask_outer : 1;
    pro.causal_rows = causal ? len : 0;
    if (inner == 1) return softmax_rows_pro(in, out, outer, len, dtype, pro, s);
    switch (dtype) {
        case MYFA_F32:  return softmax_cols_any<float, true>(static_cast<const float*>(in), static_cast<float*>(out), outer, len, inner, pro, s);
        case MYFA_F16:  return softmax_cols_any<_Float16, true>(static_cast<const _Float16*>(in), static_cast<_Float16*>(out), outer, len, inner, pro, s);
        default:        return softmax_cols_any<__bf16, true>(static_cast<const __bf16*>(in), static_cast<__bf16*>(out), outer, len, inner, pro, s);
    }
}

}  // namespace myfa
