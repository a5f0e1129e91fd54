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
//                         (256 threads; 1024 threads for columns of up to 2048 float32 / 1024 16-bit rows)
//   softmax_cols_stream   any len: online (max, sum) pass, then a pass that re-reads and writes
//                         (256 threads, or 512 / 1024 for long columns)
// VW = 4 floats or 8 halves needs inner % VW == 0 and 16-byte aligned bases; otherwise VW = 1.
#include <algorithm>
#include "softmax_common.h"
#include "tuning.h"

namespace myfa {
int softmax_rows(const void* in, void* out, int64_t h, int64_t w, int dtype, hipStream_t s);
int softmax_rows_pro(const void* in, void* out, int64_t h, int64_t w, int dtype, const Prologue& pro, hipStream_t s,
                     void* ws, size_t ws_bytes);
size_t softmax_rows_ws_bytes(int64_t h, int64_t w, int dtype);
namespace {

// Column statistics across the TY threads of a column: red is [TY][TX * VW] floats.
template <int TX, int TY, int VW, typename Op>
__device__ __forceinline__ void column_reduce(float (&v)[VW], float* red, int tx, int ty, Op op) {
    constexpr int W = TX * VW;
#pragma unroll
    for (int c = 0; c < VW; ++c) red[ty * W + c * TX + tx] = v[c];
    __syncthreads();
    // (few rows of partials: every thread adds them up itself; the unrolled reads of TY * VW values are
    // all in flight at once, so with eight-element items this form stops at eight rows: at sixteen it
    // made the 16-bit kernels 143 registers wide, three waves per SIMD)
    if constexpr (TY * VW <= 64) {
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

template <typename E, int TX, int VW, int R, bool PRO = false, int BLOCK = 256>
__global__ void __launch_bounds__(BLOCK)
softmax_cols_reg(const E* __restrict__ in, E* __restrict__ out, int64_t len, int64_t inner, int64_t nchunk, Prologue pro) {
    constexpr int TY = BLOCK / TX;
    __shared__ float red[BLOCK * VW];
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
//   PHASE 0: one workgroup per column block does both passes (grid = column blocks).
// With a workspace the rows of a column block are also split over `nsplit` workgroups, in two launches
// (grid = column blocks x nsplit, column block fastest so that workgroups running side by side
// read different columns):
//   PHASE 1: (max, sum) of the workgroup's rows -> part[block][split][2][TX * VW]; the workgroup of a
//            column block that finishes last (a counter per block, zero at launch) merges the block's
//            nsplit pairs into fin[block][2][TX * VW].  Nobody waits for anybody.
//   PHASE 2: re-reads its rows and writes, with the column block's (max, sum) from fin; the launch
//            walks the tensor from its end, starting on what phase 1 read last.
// The pairs cross workgroups inside phase 1, so they are written and read with device-scope accesses
// (past the per-XCD L2s) and need no cache-wide write-back or invalidate.
template <typename E, int TX, int VW, int BLOCK, bool PRO = false, int PHASE = 0>
__global__ void __launch_bounds__(BLOCK)
softmax_cols_stream(const E* __restrict__ in, E* __restrict__ out, int64_t len, int64_t inner, int64_t nchunk, Prologue pro,
                    int64_t groups, int nsplit, int64_t rows_per_split, unsigned* __restrict__ count, float* __restrict__ stats) {
    constexpr int TY = BLOCK / TX, U = 4, W = TX * VW;
    __shared__ float red[BLOCK * VW];
    const int tx = threadIdx.x % TX, ty = threadIdx.x / TX;
    const int64_t id = PHASE == 2 ? (int64_t)gridDim.x - 1 - blockIdx.x : blockIdx.x;
    const int64_t group = PHASE == 0 ? id : id % groups;
    const int split = PHASE == 0 ? 0 : (int)(id / groups);
    const int64_t o = group / nchunk, chunk = group % nchunk;
    const int64_t col = (chunk * TX + tx) * VW;
    const bool live = col < inner;
    const int64_t row_begin = PHASE == 0 ? 0 : split * rows_per_split;
    const int64_t row_end = PHASE == 0 ? len : (row_begin + rows_per_split < len ? row_begin + rows_per_split : len);
    const E* src = in + o * len * inner + col;
    E* dst = out + o * len * inner + col;

    float mx[VW], sum[VW], cmx[VW];
    if constexpr (PHASE != 2) {
        // pass 1: per-thread online (max, sum of 2^((x - max) * log2e)) of each of its VW columns
#pragma unroll
        for (int c = 0; c < VW; ++c) { mx[c] = -INFINITY; sum[c] = 0.f; }
        if (live) {
            for (int64_t row0 = row_begin + ty; row0 < row_end; row0 += (int64_t)U * TY) {
                float x[U][VW];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int64_t row = row0 + (int64_t)u * TY;
                    if (row < row_end) {
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
                    // branch-free online update.  While everything seen of a column is -inf the
                    // exponentials are taken against 0 instead: (-inf) - (-inf) would be NaN and survive
                    // the later rescale by 0 (leading -inf rows are ordinary input: a causal mask along
                    // this dimension); only a column that is -inf throughout comes out NaN, through
                    // 0 * inf in the second pass.  Rows past the end hold -inf and add 0.
                    const float mn = fmaxf(mx[c], cm);
                    const float base = mn == -INFINITY ? 0.f : mn;
                    float acc = sum[c] * __builtin_amdgcn_exp2f((mx[c] - base) * kLog2e);
#pragma unroll
                    for (int u = 0; u < U; ++u) acc += __builtin_amdgcn_exp2f((x[u][c] - base) * kLog2e);
                    sum[c] = acc;
                    mx[c] = mn;
                }
            }
        }
    }
    float* fin = stats + group * 2 * W + tx * VW;                         // (phases 1, 2)
    if constexpr (PHASE != 2) {
#pragma unroll
        for (int c = 0; c < VW; ++c) cmx[c] = mx[c];
        column_reduce<TX, TY, VW>(cmx, red, tx, ty, MaxOp{});
        // threads that saw no row have mx = -inf, sum = 0: they contribute 0
#pragma unroll
        for (int c = 0; c < VW; ++c)
            sum[c] = (mx[c] == -INFINITY) ? 0.f : sum[c] * __builtin_amdgcn_exp2f((mx[c] - cmx[c]) * kLog2e);
        column_reduce<TX, TY, VW>(sum, red, tx, ty, SumOp{});
    } else {
#pragma unroll
        for (int c = 0; c < VW; ++c) { cmx[c] = fin[c]; sum[c] = fin[W + c]; }
    }

    if constexpr (PHASE == 1) {
        float* part = stats + (groups + group * (int64_t)nsplit) * 2 * W + tx * VW;
        if (ty == 0) {
#pragma unroll
            for (int c = 0; c < VW; ++c) {
                put(part + (int64_t)split * 2 * W + c, cmx[c]);
                put(part + (int64_t)split * 2 * W + W + c, sum[c]);
            }
        }
        __shared__ unsigned s_last;
        // the pairs must have arrived (device-scope stores are acknowledged at the coherence point)
        // before this workgroup counts itself in: a workgroup-scope barrier alone does not wait for them
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0)
            s_last = __hip_atomic_fetch_add(count + group * kCountStride, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)(nsplit - 1);
        __syncthreads();
        if (!s_last) return;   // uniform over the workgroup
#pragma unroll
        for (int c = 0; c < VW; ++c) { mx[c] = -INFINITY; sum[c] = 0.f; }
        // The pairs are fetched with buffer loads that carry the device-scope bit (aux bit 4 = sc1):
        // coherent across the XCDs like the atomic loads they replace, but ordinary loads to the
        // compiler, which batches them across iterations (as atomics they went out one at a time, a
        // memory round trip each, on the critical path of the launch's last workgroups).
        const __amdgpu_buffer_rsrc_t prs = __builtin_amdgcn_make_buffer_rsrc((void*)(part - tx * VW), 0, (unsigned)((size_t)nsplit * 2 * W * sizeof(float)), 0x00020000);
        constexpr int kSc1 = 16;
        constexpr int MERGE_UNROLL = VW == 8 ? 1 : 4;    // (2 VW loads an iteration: sixteen are enough in flight)
#pragma unroll MERGE_UNROLL
        for (int sp = ty; sp < nsplit; sp += TY) {
            float pmv[VW], plv[VW];
#pragma unroll
            for (int c = 0; c < VW; ++c) {
                pmv[c] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(prs, (sp * 2 * W + tx * VW + c) * 4, 0, kSc1));
                plv[c] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(prs, (sp * 2 * W + W + tx * VW + c) * 4, 0, kSc1));
            }
#pragma unroll
            for (int c = 0; c < VW; ++c) {
                const float pm = pmv[c], pl = plv[c];
                const float nm = fmaxf(mx[c], pm);
                if (nm != -INFINITY)
                    sum[c] = sum[c] * __builtin_amdgcn_exp2f((mx[c] - nm) * kLog2e) + pl * __builtin_amdgcn_exp2f((pm - nm) * kLog2e);
                mx[c] = nm;
            }
        }
#pragma unroll
        for (int c = 0; c < VW; ++c) cmx[c] = mx[c];
        column_reduce<TX, TY, VW>(cmx, red, tx, ty, MaxOp{});
#pragma unroll
        for (int c = 0; c < VW; ++c)
            sum[c] = (mx[c] == -INFINITY) ? 0.f : sum[c] * __builtin_amdgcn_exp2f((mx[c] - cmx[c]) * kLog2e);
        column_reduce<TX, TY, VW>(sum, red, tx, ty, SumOp{});
        if (ty == 0) {
#pragma unroll
            for (int c = 0; c < VW; ++c) { fin[c] = cmx[c]; fin[W + c] = sum[c]; }
        }
        return;
    }

    // pass 2: re-read (what is still in L2 / Infinity Cache comes from there) and write
    if (live) {
#pragma unroll
        for (int c = 0; c < VW; ++c) sum[c] = 1.0f / sum[c];
        for (int64_t row0 = row_begin + ty; row0 < row_end; row0 += (int64_t)U * TY) {
            float x[U][VW];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int64_t row = row0 + (int64_t)u * TY;
                if (row < row_end) {
                    vload<E, VW>(x[u], src + row * inner);
                    if constexpr (PRO) prologue_col<VW>(x[u], pro, o, len, inner, row, col);
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int64_t row = row0 + (int64_t)u * TY;
                if (row < row_end) {
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
//  * Otherwise, with a workspace and too few column blocks for the chip, the rows of a column block
//    are split over workgroups as well (two launches, same traffic as the streamed form).
//  * Otherwise streamed: 1024 threads for long columns, the widest TX that leaves one workgroup
//    per CU; never narrower than 8 items (128 bytes, an L2 line) when the row has them.
// TX of the register-resident kernel for this shape, 0 when it does not apply
template <int VW>
int reg_tx(int64_t outer, int64_t len, int64_t inner, size_t esize) {
    constexpr int RMAX = 64 / VW;      // at most 64 floats of column data per thread
    const int64_t items = cdiv(inner, (int64_t)VW);
    const bool small = outer * len * inner * (int64_t)esize <= (8LL << 20);
    for (const int tx : {64, 16, 8, 4}) {
        const bool wide_enough = tx == 4 ? (items <= 4 || small) : items > tx / 2;
        const int64_t blocks = outer * cdiv(items, (int64_t)tx);
        if (wide_enough && cdiv(len, (int64_t)(256 / tx)) <= RMAX && (small || blocks >= 512 || tx == 4)) return tx;
    }
    return 0;
}

// TX of the 1024-thread register-resident kernel (16-byte items), 0 when it does not apply: the
// 256-thread one does, the column block does not fit 64 floats per thread, or there are too few
// column blocks for the chip.
template <int VW>
int reg_tx_big(int64_t outer, int64_t len, int64_t inner, size_t esize) {
    if (VW == 1 || reg_tx<VW>(outer, len, inner, esize) != 0) return 0;
    const int64_t items = cdiv(inner, (int64_t)VW);
    if (tuning().softmax_cols_big < 0) return 0;
    for (const int tx : {16, 8}) {      // (64-byte segments lose to the two-pass form: 2.7 against 3.5-3.8 TB/s at 4096 rows)
        if (items > tx / 2 && cdiv(len, (int64_t)(1024 / tx)) <= 64 / VW && outer * cdiv(items, (int64_t)tx) >= 256) return tx;
    }
    return 0;
}

// Rows split over workgroups (needs a workspace): chosen when the column blocks alone are too few
// workgroups for the chip.  tx = 0: no split.
struct SplitPlan {
    int tx = 0, nsplit = 0;
    int64_t nchunk = 0, groups = 0, rows_per_split = 0;
    size_t count_bytes = 0, bytes = 0;     // workspace: u32 count[groups], 256 bytes apart (zero at launch); float fin[groups][2][W], part[groups][nsplit][2][W]
};
constexpr int kMaxSplit = 256;

template <int VW>
SplitPlan split_plan(int64_t outer, int64_t len, int64_t inner, size_t esize) {
    SplitPlan p;
    if (reg_tx<VW>(outer, len, inner, esize) != 0) return p;
    if (reg_tx_big<VW>(outer, len, inner, esize) != 0) return p;
    const int64_t items = cdiv(inner, (int64_t)VW);
    // rows wide enough for one workgroup per CU on 128-byte segments stream at least as fast in one
    // launch (float32 [8192, 8192] over dim 0: 134 us against 140 split; float16 [4096, 32768]: 146 against 162)
    if (VW > 1 && items >= 8 && outer * cdiv(items, (int64_t)8) >= 256) return p;
    int tx = 4;
    for (const int t : {16, 8}) if (items > t / 2) { tx = t; break; }
    const int64_t groups = outer * cdiv(items, (int64_t)tx);
    int64_t target = tuning().softmax_split_wgs;
    if (target < 0) return p;
    if (target == 0) target = 1024;
    const int64_t nsplit = std::min<int64_t>({(int64_t)kMaxSplit, cdiv(target, groups), len / 64});
    if (nsplit < 2 || groups * nsplit > 0x7fffffffLL) return p;
    p.tx = tx;
    p.rows_per_split = cdiv(len, nsplit);
    p.nsplit = (int)cdiv(len, p.rows_per_split);
    p.nchunk = cdiv(items, (int64_t)tx);
    p.groups = groups;
    p.count_bytes = (size_t)groups * kCountStride * 4;
    p.bytes = p.count_bytes + (size_t)groups * (size_t)(1 + p.nsplit) * 2 * (size_t)(tx * VW) * sizeof(float);
    return p;
}

template <typename E, int VW, bool PRO>
int launch_cols(const E* in, E* out, int64_t outer, int64_t len, int64_t inner, const Prologue& pro, hipStream_t s,
                void* ws, size_t ws_bytes) {
    constexpr int RMAX = 64 / VW;
    const int64_t items = cdiv(inner, (int64_t)VW);
    auto blocks = [&](int tx) { return outer * cdiv(items, (int64_t)tx); };
    const int rtx = reg_tx<VW>(outer, len, inner, sizeof(E));
#define MYFA_REG(TXV) \
    if (rtx == TXV) { \
        const int64_t nchunk = cdiv(items, (int64_t)TXV); \
        const dim3 g((unsigned)blocks(TXV)), b(256); \
        if (cdiv(len, (int64_t)(256 / TXV)) <= RMAX / 4) launch_kernel((softmax_cols_reg<E, TXV, VW, RMAX / 4, PRO>), g, b, 0, s, in, out, len, inner, nchunk, pro); \
        else launch_kernel((softmax_cols_reg<E, TXV, VW, RMAX, PRO>), g, b, 0, s, in, out, len, inner, nchunk, pro); \
        return MYFA_OK; \
    }
    MYFA_REG(64) MYFA_REG(16) MYFA_REG(8) MYFA_REG(4)
#undef MYFA_REG
    // The same with 1024 threads, for columns four times as long: 256-byte row segments up to
    // 16 * 64 rows of float32 (8 * 64 of 16-bit elements), 128-byte ones up to twice that.
    if constexpr (VW > 1) {
        if (const int btx = reg_tx_big<VW>(outer, len, inner, sizeof(E)); btx != 0) {
            const int64_t nchunk = cdiv(items, (int64_t)btx);
            const dim3 g((unsigned)blocks(btx)), b(1024);
            if (btx == 16) launch_kernel((softmax_cols_reg<E, 16, VW, RMAX, PRO, 1024>), g, b, 0, s, in, out, len, inner, nchunk, pro);
            else           launch_kernel((softmax_cols_reg<E, 8, VW, RMAX, PRO, 1024>), g, b, 0, s, in, out, len, inner, nchunk, pro);
            return MYFA_OK;
        }
    }
    if (const SplitPlan p = split_plan<VW>(outer, len, inner, sizeof(E)); ws != nullptr && p.tx != 0 && ws_bytes >= p.bytes) {
        if (const hipError_t e = hipMemsetAsync(ws, 0, p.count_bytes, s); e != hipSuccess)
            return fail(MYFA_E_LAUNCH, "softmax: clearing the workspace: %s", hipGetErrorString(e));
        unsigned* count = static_cast<unsigned*>(ws);
        float* stats = reinterpret_cast<float*>(static_cast<char*>(ws) + p.count_bytes);
        const dim3 g((unsigned)(p.groups * p.nsplit)), b(256);
#define MYFA_SPLIT(TXV) \
        if (p.tx == TXV) { \
            launch_kernel((softmax_cols_stream<E, TXV, VW, 256, PRO, 1>), g, b, 0, s, in, out, len, inner, p.nchunk, pro, p.groups, p.nsplit, p.rows_per_split, count, stats); \
            launch_kernel((softmax_cols_stream<E, TXV, VW, 256, PRO, 2>), g, b, 0, s, in, out, len, inner, p.nchunk, pro, p.groups, p.nsplit, p.rows_per_split, count, stats); \
            return MYFA_OK; \
        }
        MYFA_SPLIT(16) MYFA_SPLIT(8) MYFA_SPLIT(4)
#undef MYFA_SPLIT
    }
    const bool big = len >= 2048;
    // (16-bit items too: 1024 threads with four rows in flight each, 1.72 -> 2.51 TB/s at bf16 8192^2 over
    // dim 0 against 512 threads with two, once the column statistics stopped costing 143 registers)
    constexpr int BIG = 1024;
#define MYFA_STREAM(TXV, LAST) \
    if (LAST || (items > TXV / 2 && blocks(TXV) >= 256)) { \
        const int64_t nchunk = cdiv(items, (int64_t)TXV); \
        const dim3 g((unsigned)blocks(TXV)); \
        if (big) launch_kernel((softmax_cols_stream<E, TXV, VW, BIG, PRO>), g, dim3(BIG), 0, s, in, out, len, inner, nchunk, pro, (int64_t)0, 1, len, (unsigned*)nullptr, (float*)nullptr); \
        else launch_kernel((softmax_cols_stream<E, TXV, VW, 256, PRO>), g, dim3(256), 0, s, in, out, len, inner, nchunk, pro, (int64_t)0, 1, len, (unsigned*)nullptr, (float*)nullptr); \
        return MYFA_OK; \
    }
    MYFA_STREAM(64, false) MYFA_STREAM(16, false) MYFA_STREAM(8, items >= 8) MYFA_STREAM(4, true)
#undef MYFA_STREAM
    return MYFA_OK;
}

template <typename E>
bool cols_vectorised(const E* in, const E* out, int64_t inner) {
    return (inner % vec_width<E>() == 0) && (reinterpret_cast<uintptr_t>(in) % 16 == 0) && (reinterpret_cast<uintptr_t>(out) % 16 == 0);
}

template <typename E, bool PRO>
int softmax_cols_any(const E* in, E* out, int64_t outer, int64_t len, int64_t inner, const Prologue& pro, hipStream_t s,
                     void* ws = nullptr, size_t ws_bytes = 0) {
    constexpr int VWV = vec_width<E>();
    const bool vec = cols_vectorised(in, out, inner);
    if (outer * cdiv(inner, (int64_t)(vec ? VWV : 1) * 4) > 0x7fffffffLL)
        return fail(MYFA_E_UNSUPPORTED, "softmax: outer=%lld x inner=%lld exceeds the grid limit", (long long)outer, (long long)inner);
    const int rc = vec ? launch_cols<E, VWV, PRO>(in, out, outer, len, inner, pro, s, ws, ws_bytes)
                       : launch_cols<E, 1, PRO>(in, out, outer, len, inner, pro, s, nullptr, 0);   // (no workspace: no split)
    if (rc != MYFA_OK) return rc;
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
size_t softmax_strided_ws_bytes(int64_t outer, int64_t len, int64_t inner, int dtype) {
    if (outer <= 0 || len <= 0 || inner <= 0) return 0;
    if (inner == 1) return softmax_rows_ws_bytes(outer, len, dtype);
    switch (dtype) {
        // (sized for 16-byte items; tensors that are not 16-byte aligned run element-wise and do not split)
        case MYFA_F32:  return inner % 4 == 0 ? split_plan<4>(outer, len, inner, 4).bytes : 0;
        case MYFA_F16:
        case MYFA_BF16: return inner % 8 == 0 ? split_plan<8>(outer, len, inner, 2).bytes : 0;
        default:        return 0;
    }
}

// ws: null, or at least softmax_strided_ws_bytes() bytes, 16-byte aligned (a smaller one is ignored)
int softmax_strided_pro(const void* in, void* out, int64_t outer, int64_t len, int64_t inner, int dtype,
                        float scale, const float* mask, int64_t mask_outer, bool causal, hipStream_t s,
                        void* ws, size_t ws_bytes) {
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
    if (inner == 1) return softmax_rows_pro(in, out, outer, len, dtype, pro, s, ws, ws_bytes);
    switch (dtype) {
        case MYFA_F32:  return softmax_cols_any<float, true>(static_cast<const float*>(in), static_cast<float*>(out), outer, len, inner, pro, s, ws, ws_bytes);
        case MYFA_F16:  return softmax_cols_any<_Float16, true>(static_cast<const _Float16*>(in), static_cast<_Float16*>(out), outer, len, inner, pro, s, ws, ws_bytes);
        default:        return softmax_cols_any<__bf16, true>(static_cast<const __bf16*>(in), static_cast<__bf16*>(out), outer, len, inner, pro, s, ws, ws_bytes);
    }
}

}  // namespace myfa
