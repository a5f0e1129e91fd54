// Fused attention forward in float32 for any sequence length and head dim <= 512.
//
//     O = softmax(scale * Q K^T) V
//
// Backs the reference's 2-D, single-head, float32 entry point
// my_attention(Q, K_t, V) (csrc/api.cpp:168-192: cuBLAS SGEMM -> my_softmax ->
// cuBLAS SGEMM, with the N x N logits and weights round-tripping through HBM) and
// the float32 flavour of the batched entry point.  Same online-softmax recurrence
// as the reference's Python Algorithm-1 restatement (tests/test_flash_attn.py:82-93).
//
// Accuracy design.  The softmax turns an absolute error in a logit into a
// relative error in the weight, and a float32 dot product of d terms of size
// |S| ~ 30-40 is off by ~1e-5 (that is the distance between the reference's own
// float32 result and the exact one).  To hold 1e-5 against an exact (float64)
// oracle the logits come from the FLOAT64 matrix instruction
// (v_mfma_f64_16x16x4_f64, inputs widened exactly), and the scaling, the
// subtraction of the row max and the change to base 2 are done in float64; the
// exponential, the row sum and the P.V product (v_mfma_f32_16x16x4_f32) are float32.
//
// Mapping.  A wave owns 16 query rows and walks 16-key tiles.  Lane l = (j, g),
// j = l & 15, g = l >> 4:
//   S^T = K Q^T   16 keys x 16 queries per tile, d/4 steps of the f64 MFMA;
//                 A = K[key j][dd], B = Q[query j][dd], dd = 16b + 4g + t (one float4 per
//                 lane and 16-wide block b).  Result c[r] = S[key 4r + g][query j]: the
//                 query sits on the lane, as in the low-precision kernel.
//   O^T += V^T P^T  the weights p[r] feed the f32 MFMA directly as its B operand
//                 (k index = g <-> key 4r + g, i.e. step r takes keys 4r..4r+3);
//                 A = V[key 4r + g][64bb + 4j + t] (float4 per lane), so accumulator
//                 4bb + t holds O[query j][64bb + 16g + 4r' + t] in register r' -- sixteen
//                 consecutive output columns per lane, stored as four float4.
// The running max is shared by the four lanes (g) of a query through two register lane
// exchanges per tile (v_permlane32_swap, v_permlane16_swap); the row sum stays a per-lane
// partial until the end.
//
// Parallelism.  A workgroup is QG x KS waves: QG groups of 16 queries that walk the
// same keys together (so K/V tiles are fetched once per workgroup as far as L1 is
// concerned) times KS interleaved shares of the key tiles, merged through LDS at
// the end.  The host picks KS > 1 only when there are too few query groups to fill
// the chip (the reference's single-head shapes).
// With a caller workspace the keys are also split over ZS workgroups (blockIdx.z): each writes
// its un-normalised partial (O, m, l) and attn_f32_merge_kernel combines them, so that the
// reference's largest test shape (1024 x 64, one head: 64 query groups) runs on every CU instead
// of 64 of them with four waves sharing each matrix pipe.
#include <algorithm>

#include "myfa_common.h"
#include "tuning.h"

namespace myfa {
namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef double f64x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float exp_or_zero(float m_old, float m_new) {
    // exp(m_old - m_new) with the convention exp(-inf - (-inf)) = 0
    return m_old == -INFINITY ? 0.f : __expf(m_old - m_new);
}

template <int NB> constexpr int max_waves() { return NB <= 4 ? 16 : NB <= 8 ? 8 : 4; }

// NB: padded head dim / 16 (4, 8, 16 or 32).  KT: K is given transposed ([d, N], the reference's
// my_attention layout) instead of [N, d].  VEC: d % 4 == 0 and all bases 16-byte aligned, so rows
// are read and written as float4; otherwise element by element.
// CAUSAL: key j is visible to query i only if j <= i (batched entry point only).
// BUF = 1 (needs VEC, K as [N, d], d == 16 NB, a head below 2 GiB, NB <= 8): K and V rows come through
// buffer loads -- rows past N read as zero by the bounds check, so the loop has no row clamp, no
// column guard and no 64-bit address arithmetic (those were 16 exec-masked branches and ~150 of
// the ~380 instructions of a tile).
// BUF = 2 (K as [d, N] -- the reference's my_attention --, a head below 2 GiB, NB <= 8, any d): the
// same for the transposed K, element by element: rows of K^T past d lie past the end of the buffer
// and read as zero; keys past N read the next row's first keys, which the softmax masks; V columns
// past d read the next key's first columns and only reach output columns that are not stored.
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
template <int NB, bool KT, bool VEC, bool CAUSAL = false, int BUF = 0>
// (256 columns, K as [N, d]: two four-wave workgroups per CU -- 235-243 registers, nothing spilled, where
// the unconstrained allocation took 300-308 and left one wave per SIMD: B2 H16 N2048 d256 29 -> 50 TFLOP/s.
// The transposed-K builds do not fit 256 registers: 46 spilled, 129 -> 175 us at 1024 x 200, so they keep one.)
__global__ void __launch_bounds__(max_waves<NB>() * 64, (NB == 16 && !KT) ? 2 : 1)
attn_fwd_f32_kernel(const float* __restrict__ Qg, const float* __restrict__ Kg,
                    const float* __restrict__ Vg, float* __restrict__ Og, float* __restrict__ LSEg,
                    int N, int d, int QG, int KS, long long head_stride, float scale,
                    int ZS, float* __restrict__ ws_o, float* __restrict__ ws_ml) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int NBB = NB / 4;                    // 64-wide column groups
    constexpr bool PREFETCH = NB <= 8;             // K and V of a tile held whole, fetched ahead of use

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int qg = wave % QG, split = wave / QG;
    const int j = lane & 15, g = lane >> 4;
    // one head is addressed with 32-bit element offsets (the host checks N * d < 2^31)
    const long long hoff = (long long)blockIdx.y * head_stride;
    const float* Qh = Qg + hoff;
    const float* Kh = Kg + hoff;
    const float* Vh = Vg + hoff;

    const int qrow = (blockIdx.x * QG + qg) * 16 + j;
    const int qrow_c = qrow < N ? qrow : N - 1;
    // causal: the wave's 16 rows see the key tiles up to their own (diagonal) tile, whose first
    // key is visible to every one of them
    const int ntiles = CAUSAL ? min((N + 15) >> 4, blockIdx.x * QG + qg + 1) : (N + 15) >> 4;
    // this workgroup's slice of the key tiles (everything unless the keys are split over blockIdx.z)
    const int tpz = (ntiles + ZS - 1) / ZS;
    const int t0 = blockIdx.z * tpz, t1 = min(ntiles, t0 + tpz);

    // four consecutive floats of a row (zero past column d)
    auto row4 = [&](const float* row, int col) __attribute__((always_inline)) -> f32x4 {
        f32x4 x = {0.f, 0.f, 0.f, 0.f};
        if constexpr (VEC) {
            if (col < d) x = *reinterpret_cast<const f32x4*>(row + col);
        } else {
#pragma unroll
            for (int t = 0; t < 4; ++t)
                if (col + t < d) x[t] = row[col + t];
        }
        return x;
    };
    // K[key][col .. col+3] in either layout
    auto k4 = [&](int key, int col) __attribute__((always_inline)) -> f32x4 {
        if constexpr (KT) {
            f32x4 x = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int t = 0; t < 4; ++t)
                if (col + t < d) x[t] = Kh[(col + t) * N + key];
            return x;
        } else {
            return row4(Kh + key * d, col);
        }
    };

    f32x4 qv[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) qv[b] = row4(Qh + qrow_c * d, 16 * b + 4 * g);

    auto visible = [&](int key) __attribute__((always_inline)) { return key < N && (!CAUSAL || key <= qrow); };

    f32x4 o[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) o[b] = f32x4{0.f, 0.f, 0.f, 0.f};
    float m = -INFINITY, l = 0.f;
    const double dscale = (double)scale;

    // (rows past N are masked in the softmax; the clamp only keeps the address valid)
    [[maybe_unused]] f32x4 kf[PREFETCH ? NB : 1];
    [[maybe_unused]] __amdgpu_buffer_rsrc_t krs, vrs;
    [[maybe_unused]] int k_voff = 0, v_voff = 0, kt_voff[4] = {0, 0, 0, 0};
    if constexpr (BUF != 0) {
        const unsigned head_bytes = (unsigned)N * (unsigned)d * 4u;
        krs = __builtin_amdgcn_make_buffer_rsrc((void*)Kh, 0, head_bytes, 0x00020000);
        vrs = __builtin_amdgcn_make_buffer_rsrc((void*)Vh, 0, head_bytes, 0x00020000);
        k_voff = (j * d + 4 * g) * 4;            // K[key0 + j][16 b + 4 g ..]: + 64 b, + key0 * d * 4
        v_voff = (g * d + 4 * j) * 4;            // V[key0 + 4 r + g][64 bb + 4 j ..]: + 256 bb, + (key0 + 4 r) * d * 4
#pragma unroll
        for (int t = 0; t < 4; ++t) kt_voff[t] = ((4 * g + t) * N + j) * 4;   // K^T[16 b + 4 g + t][key0 + j]: + (16 b N + key0) * 4
    }
    auto load_k = [&](int kt) __attribute__((always_inline)) {
        if constexpr (BUF == 1) {
#pragma unroll
            for (int b = 0; b < NB; ++b)
                kf[b] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(krs, k_voff + 64 * b, kt * 16 * d * 4, 0));
        } else if constexpr (BUF == 2) {
#pragma unroll
            for (int b = 0; b < NB; ++b)
#pragma unroll
                for (int t = 0; t < 4; ++t)
                    kf[b][t] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(krs, kt_voff[t], (16 * b * N + kt * 16) * 4, 0));
        } else {
            const int key = min(kt * 16 + j, N - 1);
#pragma unroll
            for (int b = 0; b < NB; ++b) kf[b] = k4(key, 16 * b + 4 * g);
        }
    };
    if constexpr (PREFETCH) load_k(max(min(t0 + split, ntiles - 1), 0));

    for (int kt = t0 + split; kt < t1; kt += KS) {
        const int key0 = kt * 16;
        // ---- V of this tile is requested now and consumed after the logits
        [[maybe_unused]] f32x4 vf[PREFETCH ? 4 : 1][PREFETCH ? NBB : 1];
        if constexpr (BUF == 1 || (BUF == 2 && VEC)) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int bb = 0; bb < NBB; ++bb)
                    vf[r][bb] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(vrs, v_voff + 256 * bb, (key0 + 4 * r) * d * 4, 0));
        } else if constexpr (BUF == 2) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int bb = 0; bb < NBB; ++bb)
#pragma unroll
                    for (int t = 0; t < 4; ++t)
                        vf[r][bb][t] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(vrs, v_voff + 256 * bb + 4 * t, (key0 + 4 * r) * d * 4, 0));
        } else if constexpr (PREFETCH) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float* vrow = Vh + min(key0 + 4 * r + g, N - 1) * d;
#pragma unroll
                for (int bb = 0; bb < NBB; ++bb) vf[r][bb] = row4(vrow, 64 * bb + 4 * j);
            }
        }

        // ---- S^T tile: float64 MFMA over the head dim
        f64x4 c = {0.0, 0.0, 0.0, 0.0};
        if constexpr (PREFETCH) {
            // two accumulation chains: a float64 MFMA that waits for the one before it leaves the
            // matrix pipe idle for the difference between its latency and its issue time
            f64x4 c2 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int b = 0; b < NB; ++b)
#pragma unroll
                for (int t = 0; t < 4; t += 2) {
                    c = __builtin_amdgcn_mfma_f64_16x16x4f64((double)kf[b][t], (double)qv[b][t], c, 0, 0, 0);
                    c2 = __builtin_amdgcn_mfma_f64_16x16x4f64((double)kf[b][t + 1], (double)qv[b][t + 1], c2, 0, 0, 0);
                }
            c += c2;
            // K of this wave's next tile goes into the registers just consumed; its latency
            // is covered by the softmax and the P.V product below
            load_k(BUF != 0 ? kt + KS : min(kt + KS, ntiles - 1));   // (buffer loads: a tile past the end is harmless)
        } else {
            const int key = min(key0 + j, N - 1);
#pragma unroll
            for (int bb = 0; bb < NBB; ++bb) {
                f32x4 kc[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) kc[i] = k4(key, 64 * bb + 16 * i + 4 * g);
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        float qe = qv[4 * bb + i][t];
                        // at 512 columns the widened copy of Q does not fit the register file
                        // beside Q and O: widen at each use instead of once outside the loop
                        if constexpr (NB > 16) asm volatile("" : "+v"(qe));
                        c = __builtin_amdgcn_mfma_f64_16x16x4f64((double)kc[i][t], (double)qe, c, 0, 0, 0);
                    }
                __builtin_amdgcn_sched_barrier(0);   // keep the chunks' loads from piling up in registers
            }
        }

        // ---- online softmax; c[r] is key key0 + 4r + g of query j
        double x[4];
        float tmax = -INFINITY;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            x[r] = c[r] * dscale;
            if (visible(key0 + 4 * r + g)) tmax = fmaxf(tmax, (float)x[r]);
        }
        tmax = four_rows_allreduce(tmax, MaxOp{});
        const float m_new = fmaxf(m, tmax);
        const float alpha = exp_or_zero(m, m_new);
        if (__builtin_amdgcn_ballot_w64(alpha != 1.f) != 0) {    // wave-uniform
            l *= alpha;
#pragma unroll
            for (int b = 0; b < NB; ++b) o[b] *= alpha;
        }
        m = m_new;
        float p[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float e = (float)((x[r] - (double)m) * 1.4426950408889634);
            p[r] = visible(key0 + 4 * r + g) ? __builtin_amdgcn_exp2f(e) : 0.f;
            l += p[r];
        }

        // ---- O^T += V^T P^T
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            [[maybe_unused]] const float* vrow = Vh + min(key0 + 4 * r + g, N - 1) * d;
#pragma unroll
            for (int bb = 0; bb < NBB; ++bb) {
                f32x4 v;
                if constexpr (PREFETCH) v = vf[r][bb];
                else v = row4(vrow, 64 * bb + 4 * j);
#pragma unroll
                for (int t = 0; t < 4; ++t)
                    o[4 * bb + t] = __builtin_amdgcn_mfma_f32_16x16x4f32(v[t], p[r], o[4 * bb + t], 0, 0, 0);
                if constexpr (!PREFETCH) {
                    if ((bb & 3) == 3) __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
    }

    // the four lanes of a query hold partial row sums over their own keys
    l = four_rows_allreduce(l, SumOp{});

    // ---- merge the KS key shares of each query group through LDS (share 0 collects)
    constexpr int kSlot = (NB * 4 + 2) * 64;       // floats per parked wave state
    if (KS > 1) {
        if (split > 0) {
            float* b = smem + (size_t)((split - 1) * QG + qg) * kSlot;
#pragma unroll
            for (int e = 0; e < NB; ++e)
#pragma unroll
                for (int r = 0; r < 4; ++r) b[(4 * e + r) * 64 + lane] = o[e][r];
            b[NB * 4 * 64 + lane] = m;
            b[(NB * 4 + 1) * 64 + lane] = l;
        }
        __syncthreads();
        if (split == 0) {
            for (int s = 1; s < KS; ++s) {
                const float* b = smem + (size_t)((s - 1) * QG + qg) * kSlot;
                const float m2 = b[NB * 4 * 64 + lane], l2 = b[(NB * 4 + 1) * 64 + lane];
                const float mn = fmaxf(m, m2);
                const float a1 = exp_or_zero(m, mn), a2 = exp_or_zero(m2, mn);
                l = l * a1 + l2 * a2;
#pragma unroll
                for (int e = 0; e < NB; ++e)
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[e][r] = o[e][r] * a1 + b[(4 * e + r) * 64 + lane] * a2;
                m = mn;
            }
        }
    }
    if (split != 0 || qrow >= N) return;

    // ---- store: accumulator 4bb + t, register r holds column 64bb + 16g + 4r + t
    // (keys split over workgroups: the un-normalised partial and its (m, l) go to the workspace)
    const bool partial = ZS > 1;
    const float inv = partial ? 1.0f : 1.0f / l;
    const long long prow = ((long long)blockIdx.y * ZS + blockIdx.z) * N + qrow;
    float* orow = partial ? ws_o + prow * d : Og + hoff + qrow * d;
    if (partial && g == 0) { ws_ml[2 * prow] = m; ws_ml[2 * prow + 1] = l; }
#pragma unroll
    for (int bb = 0; bb < NBB; ++bb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int col = 64 * bb + 16 * g + 4 * r;
            const f32x4 w = {o[4 * bb][r] * inv, o[4 * bb + 1][r] * inv, o[4 * bb + 2][r] * inv, o[4 * bb + 3][r] * inv};
            if constexpr (VEC) {
                if (col < d) *reinterpret_cast<f32x4*>(orow + col) = w;
            } else {
#pragma unroll
                for (int t = 0; t < 4; ++t)
                    if (col + t < d) orow[col + t] = w[t];
            }
        }
    if (!partial && LSEg != nullptr && g == 0) LSEg[(long long)blockIdx.y * N + qrow] = m + logf(l);
}

// O[row][col] = sum_s e^(m_s - M) O_s[row][col] / sum_s e^(m_s - M) l_s over the ZS key slices
// (M = max_s m_s; a slice without keys has m = -inf, l = 0 and drops out).
__global__ void __launch_bounds__(256)
attn_f32_merge_kernel(const float* __restrict__ ws_o, const float* __restrict__ ws_ml, float* __restrict__ Og,
                      float* __restrict__ LSEg, int N, int d, int ZS) {
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    if (e >= (long long)N * d) return;
    const int row = (int)(e / d), col = (int)(e - (long long)row * d);
    const long long head = blockIdx.y;
    float M = -INFINITY;
    for (int s = 0; s < ZS; ++s) M = fmaxf(M, ws_ml[2 * ((head * ZS + s) * N + row)]);
    float num = 0.f, den = 0.f;
    for (int s = 0; s < ZS; ++s) {
        const long long pr = (head * ZS + s) * N + row;
        const float a = exp_or_zero(ws_ml[2 * pr], M);
        num += a * ws_o[pr * d + col];
        den += a * ws_ml[2 * pr + 1];
    }
    Og[(head * N + row) * d + col] = num / den;
    if (LSEg != nullptr && col == 0) LSEg[head * N + row] = M + logf(den);
}

// Key slices over workgroups for a problem with `wgs` workgroups of its own: only when those
// leave most of the chip idle and every wave still gets two key tiles; a power of two.
inline int key_slices(int64_t wgs, int64_t ntiles, int ks_in) {
    const int target = tuning().attn_f32_wgs > 0 ? tuning().attn_f32_wgs : 256;
    int zs = 1;
    while (wgs * zs * 2 <= target && ntiles / ((int64_t)zs * 2 * ks_in) >= 2) zs *= 2;
    return zs;
}

template <int NB, bool KT, bool VEC, bool CAUSAL = false>
int launch(const float* q, const float* k, const float* v, float* o, float* lse,
           int64_t heads, int64_t N, int64_t d, float scale, hipStream_t s,
           float* ws = nullptr, size_t ws_bytes = 0) {
    constexpr bool can_fast = VEC && !KT && NB <= 8;
    constexpr bool can_kt = KT && !CAUSAL && NB <= 8;
    const bool small_head = N * d * 4 < (1LL << 31);
    auto kern = attn_fwd_f32_kernel<NB, KT, VEC, CAUSAL, 0>;
    if constexpr (can_fast) { if (small_head && d == 16 * NB) kern = attn_fwd_f32_kernel<NB, KT, VEC, CAUSAL, 1>; }
    if constexpr (can_kt) { if (small_head) kern = attn_fwd_f32_kernel<NB, KT, VEC, CAUSAL, 2>; }
    constexpr int wmax = max_waves<NB>();
    const int64_t qgroups = cdiv(N, 16);
    const int64_t ntiles = cdiv(N, 16);
    // Key shares only when the query groups alone cannot fill the chip (2 waves per SIMD of
    // 256 CUs = 2048 waves); then as many as fit the workgroup, a power of two, at most one
    // per key tile.
    int ks = 1;
    while (ks < wmax && ks * 2 <= ntiles && qgroups * heads * ks < 2048) ks *= 2;
    // Query groups per workgroup share their K/V fetches; keep at least ~2 workgroups per CU.
    int qgn = wmax / ks;
    while (qgn > 1 && cdiv(qgroups, qgn) * heads < 512) qgn /= 2;
    // With a workspace: four key shares inside the workgroup (one wave per SIMD) and the rest of
    // the split over workgroups, as far as key_slices() allows.
    int zs = 1;
    if (!CAUSAL && ws != nullptr && qgn == 1 && ks > 4) {
        zs = key_slices(qgroups * heads, ntiles, 4);
        if (zs > 1 && (size_t)heads * zs * N * (d + 2) * sizeof(float) <= ws_bytes) ks = 4;
        else zs = 1;
    }
    const size_t lds_bytes = ks > 1 ? (size_t)(ks - 1) * qgn * (NB * 4 + 2) * 64 * sizeof(float) : 0;
    if (lds_bytes > 160 * 1024)
        return fail(MYFA_E_UNSUPPORTED, "attention (f32): merge buffer of %zu bytes exceeds LDS", lds_bytes);
    if (lds_bytes > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (e != hipSuccess)
            return fail(MYFA_E_LAUNCH, "attention (f32): cannot reserve %zu bytes of LDS: %s", lds_bytes, hipGetErrorString(e));
    }
    const dim3 grid((unsigned)cdiv(qgroups, qgn), (unsigned)heads, (unsigned)zs);
    float* ws_o = ws;
    float* ws_ml = zs > 1 ? ws + (size_t)heads * zs * N * d : nullptr;
    launch_kernel(kern, grid, dim3(qgn * ks * 64), lds_bytes, s, q, k, v, o, lse, (int)N, (int)d,
                       qgn, ks, (long long)(N * d), scale, zs, ws_o, ws_ml);
    if (zs > 1)
        launch_kernel(attn_f32_merge_kernel, dim3((unsigned)cdiv(N * d, 256), (unsigned)heads), dim3(256), 0, s,
                      ws_o, ws_ml, o, lse, (int)N, (int)d, zs);
    return check_launch("attn_fwd_f32");
}

template <int NB>
int launch_layout(const float* q, const float* k, const float* v, float* o, float* lse,
                  int64_t heads, int64_t N, int64_t d, bool kt, bool vec, bool causal, float scale, hipStream_t s,
                  float* ws, size_t ws_bytes) {
    if (causal) {
        if (kt) return fail(MYFA_E_UNSUPPORTED, "attention (f32): the causal mask needs K as [N, d]");
        return vec ? launch<NB, false, true, true>(q, k, v, o, lse, heads, N, d, scale, s)
                   : launch<NB, false, false, true>(q, k, v, o, lse, heads, N, d, scale, s);
    }
    if (kt) return vec ? launch<NB, true, true>(q, k, v, o, lse, heads, N, d, scale, s, ws, ws_bytes)
                       : launch<NB, true, false>(q, k, v, o, lse, heads, N, d, scale, s, ws, ws_bytes);
    return vec ? launch<NB, false, true>(q, k, v, o, lse, heads, N, d, scale, s, ws, ws_bytes)
               : launch<NB, false, false>(q, k, v, o, lse, heads, N, d, scale, s, ws, ws_bytes);
}

}  // namespace

// K element (key j, dim c) of a head is at k[j * k_key_stride + c * k_d_stride]:
//   K_t layout [d, N] (the reference's my_attention): k_key_stride = 1, k_d_stride = N
//   K   layout [N, d] (batched entry point):          k_key_stride = d, k_d_stride = 1
// `ws` (may be null): float workspace of ws_bytes for key slices over workgroups, see
// attn_fwd_f32_ws_bytes().
int attn_fwd_f32(const float* q, const float* k, const float* v, float* o, float* lse,
                 int64_t heads, int64_t N, int64_t d, int64_t k_key_stride, int64_t k_d_stride,
                 float scale, bool causal, hipStream_t s, float* ws, size_t ws_bytes) {
    if (d > 512) return fail(MYFA_E_UNSUPPORTED, "attention (f32): head dim %lld > 512", (long long)d);
    if (N * d >= (1LL << 31))
        return fail(MYFA_E_UNSUPPORTED, "attention (f32): N*d=%lld per head beyond the kernel's limit of 2^31", (long long)(N * d));
    // heads ride on the grid's y extent: more than it holds go in successive launches on the stream
    constexpr int64_t kMaxHeads = 65535;
    if (heads > kMaxHeads) {
        for (int64_t h0 = 0; h0 < heads; h0 += kMaxHeads) {
            const int64_t off = h0 * N * d;
            const int rc = attn_fwd_f32(q + off, k + off, v + off, o + off, lse ? lse + h0 * N : nullptr, std::min(kMaxHeads, heads - h0),
                                        N, d, k_key_stride, k_d_stride, scale, causal, s, ws, ws_bytes);
            if (rc != MYFA_OK) return rc;
        }
        return MYFA_OK;
    }
    // (with d == 1 or N == 1 the two layouts are the same bytes)
    const bool rowmajor = (k_key_stride == d && k_d_stride == 1) || d == 1 || N == 1;
    const bool kt = !rowmajor;
    if (kt && !(k_key_stride == 1 && k_d_stride == N))
        return fail(MYFA_E_UNSUPPORTED, "attention (f32): K must be [N, d] or [d, N] contiguous");
    const bool vec = d % 4 == 0 &&
        ((reinterpret_cast<uintptr_t>(q) | reinterpret_cast<uintptr_t>(k) | reinterpret_cast<uintptr_t>(v) |
          reinterpret_cast<uintptr_t>(o)) % 16 == 0);
    if (d <= 64)  return launch_layout<4>(q, k, v, o, lse, heads, N, d, kt, vec, causal, scale, s, ws, ws_bytes);
    if (d <= 128) return launch_layout<8>(q, k, v, o, lse, heads, N, d, kt, vec, causal, scale, s, ws, ws_bytes);
    if (d <= 256) return launch_layout<16>(q, k, v, o, lse, heads, N, d, kt, vec, causal, scale, s, ws, ws_bytes);
    return launch_layout<32>(q, k, v, o, lse, heads, N, d, kt, vec, causal, scale, s, ws, ws_bytes);
}

// Bytes of workspace with which attn_fwd_f32 can spread `heads` problems of N keys over the
// whole chip; 0 when it would not split the keys over workgroups anyway.
size_t attn_fwd_f32_ws_bytes(int64_t heads, int64_t N, int64_t d) {
    if (N <= 0 || d <= 0 || d > 512 || heads <= 0) return 0;
    const int64_t qgroups = cdiv(N, 16);
    const int zs = key_slices(qgroups * heads, qgroups, 4);
    return zs > 1 ? (size_t)heads * zs * N * (d + 2) * sizeof(float) : 0;
}

}  // namespace myfa
