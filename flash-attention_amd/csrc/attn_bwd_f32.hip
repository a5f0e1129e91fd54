// Float32 backward of the attention forward, on the float32 matrix instruction
// (v_mfma_f32_16x16x4_f32: 157 TFLOP/s dense on MI355X).
//
//   P  = exp(scale * Q K^T - lse)            (recomputed from the forward's row log-sum-exp)
//   dV = P^T dO
//   dP = dO V^T
//   dS = P o (dP - delta),  delta[q] = sum_c dO[q][c] O[q][c]
//   dQ = scale * dS K,      dK = scale * dS^T Q
//
// No counterpart in the reference, which has neither a fused forward nor a backward
// (csrc/my_flash_attn/flash_attention.cu:23-69 is a stub).
//
// One kernel template in two roles.  A wave OWNS 16 rows of one side and streams 16-row blocks of
// the other side through LDS (a workgroup = 4 waves = 64 owned rows, sharing the streamed blocks):
//
//   role dK/dV: owns keys    (K, V fragments in registers), streams queries (Q, dO blocks in LDS)
//   role dQ:    owns queries (Q, dO fragments in registers), streams keys   (K, V blocks in LDS)
//
// Both roles form the 16 x 16 block  S[streamed row][owned column]  with the streamed side as the
// A operand.  The instruction leaves row 4g + r, column j in register r of lane 16 g + j -- which is
// exactly where the A operand of the NEXT product wants it: the products that consume P or dS
// contract over the streamed rows, A[i = owned column j][k = g] = P[4 g + r][j] is register r as it
// stands, with B[k = g][j = c] = Y[4 g + r][16 cb + c] read from the LDS block.  (The contraction
// index runs over {r, 4 + r, 8 + r, 12 + r} for r = 0..3 instead of 0..15 in order; a sum does not
// care.)  So no value is ever transposed, through LDS or lanes.
//
// The first two products read the streamed block as float4 per lane (columns 16 t + 4 g .. + 3 of
// row j, against the same four columns of the owned row held in registers): the contraction runs
// over {4 g + e} for e = 0..3 per step, again a permutation.
//
// LDS rows are padded by 4 floats: the 16 lanes of one float4 read then cover all 64 banks once,
// and so do the 64 lanes of one scalar read of the second kind (rows 4 g + r, 16 columns each).
#include <algorithm>

#include "myfa_common.h"

namespace myfa {

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr float kLog2e = 1.4426950408889634f;

// delta[row] = sum_c dO[row][c] * O[row][c]: 16 lanes per row
__global__ void __launch_bounds__(256)
attn_bwd_f32_delta_kernel(const float* __restrict__ dO, const float* __restrict__ O, float* __restrict__ delta,
                          long long rows, int d) {
    const long long row = (long long)blockIdx.x * 16 + (threadIdx.x >> 4);
    const int sub = threadIdx.x & 15;
    float acc = 0.f;
    if (row < rows)
        for (int c = sub; c < d; c += 16) acc += dO[row * d + c] * O[row * d + c];
    acc += dpp_mov<0xB1>(acc);    // quad_perm [1,0,3,2]
    acc += dpp_mov<0x4E>(acc);    // quad_perm [2,3,0,1]
    acc += dpp_mov<0x141>(acc);   // row_half_mirror
    acc += dpp_mov<0x140>(acc);   // row_mirror
    if (row < rows && sub == 0) delta[row] = acc;
}

// two buffers of two streamed blocks (rows padded by 4 floats) and of their 16 + 16 row statistics
template <int DPAD>
constexpr size_t lds_floats() { return 4 * 16 * (DPAD + 4) + 64; }

// DPAD: head dim rounded up to a multiple of 16 the build covers; DQ: the role (see above);
// VEC: d % 4 == 0 (rows are 16-byte aligned: float4 global loads).
template <int DPAD, bool DQ, bool CAUSAL, bool VEC>
__global__ void __launch_bounds__(256)
attn_bwd_f32_kernel(const float* __restrict__ Qg, const float* __restrict__ Kg, const float* __restrict__ Vg,
                    const float* __restrict__ dOg, const float* __restrict__ LSEg, const float* __restrict__ Dg,
                    float* __restrict__ out1, float* __restrict__ out2, int heads, int N, int d, float scale) {
    constexpr int NT = DPAD / 16;              // float4 steps of the first products / 16-column blocks of the outputs
    constexpr int LD = DPAD + 4;               // LDS row stride in floats
    constexpr int CH = 16 * DPAD / 4;          // float4 chunks of one streamed block
    constexpr int CPT = (CH + 255) / 256;      // ... per thread
    extern __shared__ __attribute__((aligned(16))) float smem[];   // lds_floats<DPAD>() of them
    float (*sY1)[16 * LD] = reinterpret_cast<float (*)[16 * LD]>(smem);
    float (*sY2)[16 * LD] = reinterpret_cast<float (*)[16 * LD]>(smem + 2 * 16 * LD);
    float (*sL)[16] = reinterpret_cast<float (*)[16]>(smem + 4 * 16 * LD);
    float (*sD)[16] = reinterpret_cast<float (*)[16]>(smem + 4 * 16 * LD + 32);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 15, g = lane >> 4;
    // Workgroups go to the 8 XCDs round robin in dispatch order (x fastest).  All 64-row blocks of a
    // head are given to ONE XCD, consecutively: they stream the same blocks of the other side at about
    // the same time, so that XCD's L2 fetches each block once.  (The grid's y extent is the head
    // count rounded up to a multiple of 8.)
    const int nxb = gridDim.x;
    const long long lin = (long long)blockIdx.y * nxb + blockIdx.x;
    const long long slot = lin >> 3;
    const long long head = (slot / nxb) * 8 + (lin & 7);
    if (head >= heads) return;                 // (the whole workgroup, in front of every barrier)
    int xb = (int)(slot % nxb);
    // causal, role dQ: the late query blocks see the most keys -- longest first
    if constexpr (CAUSAL && DQ) xb = nxb - 1 - xb;
    const long long hoff = head * (long long)N * d;
    const float* X1 = (DQ ? Qg : Kg) + hoff;   // owned side: feeds S
    const float* X2 = (DQ ? dOg : Vg) + hoff;  //             feeds dP
    const float* Y1 = (DQ ? Kg : Qg) + hoff;   // streamed side: feeds S, and the output that takes dS
    const float* Y2 = (DQ ? Vg : dOg) + hoff;  //                feeds dP, and (role dK/dV) the output that takes P
    const float* lse = LSEg + head * N;
    const float* dlt = Dg + head * N;
    const float sl2 = scale * kLog2e;

    const int own0 = xb * 64 + wave * 16;                   // this wave's owned rows
    const int orow = own0 + j;

    auto row4 = [&](const float* base, int row, int col) __attribute__((always_inline)) -> f32x4 {
        f32x4 x = {0.f, 0.f, 0.f, 0.f};
        if (row < N) {
            if constexpr (VEC) {
                if (col < d) x = *reinterpret_cast<const f32x4*>(base + (long long)row * d + col);
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (col + e < d) x[e] = base[(long long)row * d + col + e];
            }
        }
        return x;
    };

    // owned fragments: row j, columns 16 t + 4 g .. + 3
    f32x4 x1[NT], x2[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        x1[t] = row4(X1, orow, 16 * t + 4 * g);
        x2[t] = row4(X2, orow, 16 * t + 4 * g);
    }
    // role dQ: the owned column is a query, its statistics are fixed
    [[maybe_unused]] float own_l2 = INFINITY, own_dl = 0.f;
    if constexpr (DQ) {
        if (orow < N) { own_l2 = lse[orow] * kLog2e; own_dl = dlt[orow]; }
    }

    // Above 256 columns the owned fragments alone are half the register file: the output columns are
    // then taken CS groups of NTC blocks at a time, each group in a walk of its own over the streamed
    // side (S and dP are formed again in every walk).
    constexpr int CS = DPAD > 256 ? (DQ ? 2 : 4) : 1;
    constexpr int NTC = NT / CS;
    f32x4 acc1[NTC], acc2[NTC];                // role dK/dV: dV, dK; role dQ: acc2 = dQ only

    // streamed blocks this workgroup needs
    const int nblk = (N + 15) >> 4;
    int sb0 = 0, sb1 = nblk;
    if constexpr (CAUSAL) {
        if constexpr (DQ) sb1 = min(nblk, xb * 4 + 4);                  // keys up to the last owned query
        else sb0 = xb * 4;                                              // queries from the first owned key on
    }
    // Causal, role dK/dV: walk the queries from the last block down, so that the workgroups of a head
    // that start together read the same block at the same time (role dQ: all start at key block 0).
    constexpr bool REV = CAUSAL && !DQ;
    const int nsb = sb1 - sb0;
    auto block_at = [&](int it) __attribute__((always_inline)) { return REV ? sb1 - 1 - it : sb0 + it; };

    f32x4 st1[CPT], st2[CPT];
    float stl = INFINITY, std_ = 0.f;
    auto fetch = [&](int sb) __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < CPT; ++u) {
            const int c = tid + 256 * u;
            const int row = sb * 16 + c / (DPAD / 4), col = 4 * (c % (DPAD / 4));
            if (CH % 256 == 0 || c < CH) {
                st1[u] = row4(Y1, row, col);
                st2[u] = row4(Y2, row, col);
            }
        }
        if constexpr (!DQ) {
            // the streamed rows are queries: their statistics travel with the block
            if (tid < 16) {
                const int row = sb * 16 + tid;
                stl = row < N ? lse[row] * kLog2e : INFINITY;
                std_ = row < N ? dlt[row] : 0.f;
            }
        }
    };
    auto park = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < CPT; ++u) {
            const int c = tid + 256 * u;
            const int off = (c / (DPAD / 4)) * LD + 4 * (c % (DPAD / 4));
            if (CH % 256 == 0 || c < CH) {
                *reinterpret_cast<f32x4*>(&sY1[buf][off]) = st1[u];
                *reinterpret_cast<f32x4*>(&sY2[buf][off]) = st2[u];
            }
        }
        if constexpr (!DQ) {
            if (tid < 16) { sL[buf][tid] = stl; sD[buf][tid] = std_; }
        }
    };

    float* o1 = out1 != nullptr ? out1 + hoff : nullptr;
    float* o2 = out2 + hoff;
#pragma nounroll
    for (int cs = 0; cs < CS; ++cs) {
#pragma unroll
    for (int t = 0; t < NTC; ++t) acc1[t] = acc2[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (nsb > 0) {
        fetch(block_at(0));
        park(0);                               // (a second walk: everybody left both buffers at the last barrier)
    }
    __syncthreads();

    for (int it = 0; it < nsb; ++it) {
        const int sb = block_at(it);
        const int buf = it & 1;
        if (it + 1 < nsb) fetch(block_at(it + 1));   // in flight under this block's matrix work

        const float* y1 = sY1[buf];
        const float* y2 = sY2[buf];
        f32x4 s = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const f32x4 a1 = *reinterpret_cast<const f32x4*>(y1 + j * LD + 16 * t + 4 * g);
            const f32x4 a2 = *reinterpret_cast<const f32x4*>(y2 + j * LD + 16 * t + 4 * g);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                s = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[e], x1[t][e], s, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_16x16x4f32(a2[e], x2[t][e], dp, 0, 0, 0);
            }
        }
        // register r: streamed row 4 g + r, owned column j
        float p[4], ds[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int srow = sb * 16 + 4 * g + r;
            const int q = DQ ? orow : srow, key = DQ ? srow : orow;
            float l2, dl;
            if constexpr (DQ) { l2 = own_l2; dl = own_dl; }
            else { l2 = sL[buf][4 * g + r]; dl = sD[buf][4 * g + r]; }
            const bool ok = key < N && q < N && (!CAUSAL || key <= q);
            p[r] = ok ? __builtin_amdgcn_exp2f(s[r] * sl2 - l2) : 0.f;
            ds[r] = p[r] * (dp[r] - dl);
        }
        const int c0 = 16 * NTC * cs + j;      // this walk's first output column of the lane
#pragma unroll
        for (int cb = 0; cb < NTC; ++cb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float b1 = y1[(4 * g + r) * LD + 16 * cb + c0];
                acc2[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(ds[r], b1, acc2[cb], 0, 0, 0);
                if constexpr (!DQ) {
                    const float b2 = y2[(4 * g + r) * LD + 16 * cb + c0];
                    acc1[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(p[r], b2, acc1[cb], 0, 0, 0);
                }
            }

        if (it + 1 < nsb) park(buf ^ 1);       // the other buffer was last read one iteration ago, behind a barrier
        __syncthreads();
    }

    // ---- store: accumulator cb, register r = owned row 4 g + r, column 16 (NTC cs + cb) + j
#pragma unroll
    for (int cb = 0; cb < NTC; ++cb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = own0 + 4 * g + r, col = 16 * (NTC * cs + cb) + j;
            if (row < N && col < d) {
                o2[(long long)row * d + col] = acc2[cb][r] * scale;
                if constexpr (!DQ) o1[(long long)row * d + col] = acc1[cb][r];
            }
        }
    }
}

template <int DPAD, bool CAUSAL, bool VEC>
int launch_f32(const float* q, const float* k, const float* v, const float* o, const float* lse, const float* dout,
               float* dq, float* dk, float* dv, float* delta, int64_t heads, int64_t N, int64_t d, float scale, hipStream_t s) {
    launch_kernel(attn_bwd_f32_delta_kernel, dim3((unsigned)cdiv(heads * N, 16)), dim3(256), 0, s, dout, o, delta,
                  (long long)(heads * N), (int)d);
    const dim3 grid((unsigned)cdiv(N, 64), (unsigned)(cdiv(heads, 8) * 8));   // (see the kernel's head mapping)
    constexpr size_t lds = lds_floats<DPAD>() * sizeof(float);
    if constexpr (lds > 64 * 1024) {
        for (const void* kern : {reinterpret_cast<const void*>(attn_bwd_f32_kernel<DPAD, false, CAUSAL, VEC>),
                                 reinterpret_cast<const void*>(attn_bwd_f32_kernel<DPAD, true, CAUSAL, VEC>)}) {
            const hipError_t e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess)
                return fail(MYFA_E_LAUNCH, "attention backward (float32): cannot reserve %zu bytes of LDS: %s", lds, hipGetErrorString(e));
        }
    }
    launch_kernel((attn_bwd_f32_kernel<DPAD, false, CAUSAL, VEC>), grid, dim3(256), lds, s, q, k, v, dout, lse, (const float*)delta,
                  dv, dk, (int)heads, (int)N, (int)d, scale);
    launch_kernel((attn_bwd_f32_kernel<DPAD, true, CAUSAL, VEC>), grid, dim3(256), lds, s, q, k, v, dout, lse, (const float*)delta,
                  (float*)nullptr, dq, (int)heads, (int)N, (int)d, scale);
    return check_launch("attn_bwd_f32");
}

template <int DPAD>
int launch_f32_flags(const float* q, const float* k, const float* v, const float* o, const float* lse, const float* dout,
                     float* dq, float* dk, float* dv, float* delta, int64_t heads, int64_t N, int64_t d, float scale,
                     bool causal, hipStream_t s) {
    const bool vec = d % 4 == 0;
    if (causal) return vec ? launch_f32<DPAD, true, true>(q, k, v, o, lse, dout, dq, dk, dv, delta, heads, N, d, scale, s)
                           : launch_f32<DPAD, true, false>(q, k, v, o, lse, dout, dq, dk, dv, delta, heads, N, d, scale, s);
    return vec ? launch_f32<DPAD, false, true>(q, k, v, o, lse, dout, dq, dk, dv, delta, heads, N, d, scale, s)
               : launch_f32<DPAD, false, false>(q, k, v, o, lse, dout, dq, dk, dv, delta, heads, N, d, scale, s);
}

}  // namespace

// Float32 tensors [heads, N, d], any d up to 512 (the float32 forward's range), any N; `delta`: float32 workspace of at least
// heads * N elements (the caller's; the library allocates nothing).
int attn_bwd_f32(const float* q, const float* k, const float* v, const float* o, const float* lse, const float* dout,
                 float* dq, float* dk, float* dv, float* delta, int64_t heads, int64_t N, int64_t d,
                 float scale, bool causal, hipStream_t s) {
    if (d > 512) return fail(MYFA_E_UNSUPPORTED, "attention backward (float32): head dim %lld > 512", (long long)d);
    if (N * d >= (1LL << 31))
        return fail(MYFA_E_UNSUPPORTED, "attention backward (float32): N*d=%lld per head beyond the kernel's limit of 2^31", (long long)(N * d));
    // heads ride on the grid's y extent (rounded up to a multiple of 8): more go in successive launches
    constexpr int64_t kMaxHeads = 65528;
    if (heads > kMaxHeads) {
        for (int64_t h0 = 0; h0 < heads; h0 += kMaxHeads) {
            const int64_t off = h0 * N * d;
            const int rc = attn_bwd_f32(q + off, k + off, v + off, o + off, lse + h0 * N, dout + off, dq + off, dk + off, dv + off,
                                        delta + h0 * N, std::min(kMaxHeads, heads - h0), N, d, scale, causal, s);
            if (rc != MYFA_OK) return rc;
        }
        return MYFA_OK;
    }
    if (d <= 32)  return launch_f32_flags<32>(q, k, v, o, lse, dout, dq, dk, dv, delta, heads, N, d, scale, causal, s);
    if (d <= 64)  return launch_f32_flags<64>(q, k, v, o, lse, dout, dq, dk, dv, delta, heads, N, d, scale, causal, s);
    if (d <= 128) return launch_f32_flags<128>(q, k, v, o, lse, dout, dq, dk, dv, delta, heads, N, d, scale, causal, s);
    if (d <= 256) return launch_f32_flags<256>(q, k, v, o, lse, dout, dq, dk, dv, delta, heads, N, d, scale, causal, s);
    return launch_f32_flags<512>(q, k, v, o, lse, dout, dq, dk, dv, delta, heads, N, d, scale, causal, s);
}

}  // namespace myfa
