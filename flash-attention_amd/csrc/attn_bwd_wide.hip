// Attention backward for f16 / bf16 head dims above 128 (192 and 256) on gfx950 matrix cores.
//
// Same formulation as attn_bwd_mfma.hip (dV = P^T dO, dS = P o (dO V^T - delta), dQ = scale dS K,
// dK = scale dS^T Q, P recomputed from the saved row statistic), for rows too wide for that
// file's register plan.  What changes with the width:
//   * a workgroup is four waves, one per SIMD, so that a wave may use all 512 registers of a lane:
//     a 32-row operand held as MFMA fragments is d/4 registers (64 at d = 256) and a 32 x d float
//     accumulator d/2 (128);
//   * streamed tiles are 32 rows (16 KiB at d = 256), four LDS buffers per streamed tensor (the tile in
//     use, the next one complete, two more in flight behind a counted wait);
//   * S and dP are VGPR-form MFMAs and the gradient tile is pinned to accumulator registers (at this
//     pressure the compiler otherwise parks every MFMA result in accumulator registers and moves S
//     and dP out and back for the vector work);
//   * dK and dV are two launches of one kernel (each holds ONE d-wide accumulator next to the K, V
//     fragments; both passes form S, five tile products per pair of passes instead of four).
// The reference has no backward (csrc/my_flash_attn/flash_attention.cu:23-69).
#include "attn_mfma_common.h"

namespace myfa {
namespace {

constexpr float kLog2eW = 1.4426950408889634f;
constexpr int kRowsW = 32;           // rows of a streamed tile
constexpr int kThreadsW = 256;

// -delta = -rowsum(dO o O) and lse * log2(e) per row: [-delta | lse2], 2 * rows floats.
// 16 lanes per row, each over d / 128 (rounded up) 16-byte chunks.
template <typename T, int D>
__global__ void __launch_bounds__(256)
attn_bwd_wide_delta_kernel(const T* __restrict__ Og, const T* __restrict__ dOg, const float* __restrict__ LSEg,
                           float* __restrict__ DELg, long long rows) {
    using vec8 = typename Mma<T>::vec8;
    const long long row = (long long)blockIdx.x * 16 + threadIdx.x / 16;
    const int ch = threadIdx.x % 16;
    float acc = 0.f;
    if (row < rows) {
#pragma unroll
        for (int c = ch; c < D / 8; c += 16) {
            const vec8 a = *reinterpret_cast<const vec8*>(Og + row * D + c * 8);
            const vec8 b = *reinterpret_cast<const vec8*>(dOg + row * D + c * 8);
#pragma unroll
            for (int j = 0; j < 8; ++j) acc += (float)a[j] * (float)b[j];
        }
    }
    acc += dpp_mov<0xB1>(acc);      // lanes ^ 1, ^ 2 inside quads, the mirrored quad, the mirrored half row
    acc += dpp_mov<0x4E>(acc);
    acc += dpp_mov<0x141>(acc);
    acc += dpp_mov<0x140>(acc);
    if (row < rows && ch == 0) {
        DELg[row] = -acc;
        DELg[rows + row] = LSEg[row] * kLog2eW;
    }
}

// Fragment addressing shared by the two kernels (the tile image of attn_mfma_common.h):
//   row read, k-step ks:            rbase[ks & 1] + 512 (ks >> 1)             (row = lane & 31)
//   transposing read of (st, db):   tbase[0 | 1] + 16 D * 2 st + 512 db        (16 streamed rows x 32 columns)
template <int D>
struct FragBase {
    int rbase[2], tbase[2];
    __device__ __forceinline__ explicit FragBase(int lane) {
        const int r = lane & 31, h = lane >> 5;
        const int tq = (lane & 15) >> 2, tp = lane & 3, tg1 = (lane >> 4) & 1;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            rbase[e] = 16 * D * (r >> 3) + 64 * (r & 7) + 16 * ((2 * e + h) ^ ((r >> 2) & 3));
            tbase[e] = 64 * (4 * h + tq) + 16 * ((2 * tg1 + (tp >> 1)) ^ (2 * e + h)) + 8 * (tp & 1) + 16 * D * e;
        }
    }
    __device__ __forceinline__ s16x8 row(const char* tile, int ks) const {
        return *reinterpret_cast<const s16x8*>(tile + rbase[ks & 1] + 512 * (ks >> 1));
    }
    // The same two reads as instructions the compiler does not see as LDS reads: hipcc makes an LDS
    // read that follows an LDS-DMA fetch wait for ALL fetches in flight (it cannot tell the buffers
    // apart), and the kernels below keep the fetches of two tiles in flight.  `tile` is the buffer's
    // LDS byte address; the caller waits for the fragment with frag_wait<younger requests>() before
    // its first use.  What orders these reads against the fetches: a buffer is fetched into only
    // behind the barrier that follows its last read, and read only behind the barrier that follows
    // the fetching waves' counted s_waitcnt vmcnt.
    template <int KSTEP>
    __device__ __forceinline__ s16x8 row_a([[maybe_unused]] unsigned tile) const {
        s16x8 a = {};
#if defined(__HIP_DEVICE_COMPILE__)
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(a) : "v"(tile + rbase[KSTEP & 1]), "n"(512 * (KSTEP >> 1)));
#endif
        return a;
    }
    template <int ST, int DBLK>
    __device__ __forceinline__ s16x8 tr_a([[maybe_unused]] unsigned tile) const {
        s16x4 lo = {}, hi = {};
#if defined(__HIP_DEVICE_COMPILE__)
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(lo) : "v"(tile + tbase[0]), "n"(16 * D * 2 * ST + 512 * DBLK));
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(hi) : "v"(tile + tbase[1]), "n"(16 * D * 2 * ST + 512 * DBLK));
#endif
        s16x8 a;
        a[0] = lo[0]; a[1] = lo[1]; a[2] = lo[2]; a[3] = lo[3];
        a[4] = hi[0]; a[5] = hi[1]; a[6] = hi[2]; a[7] = hi[3];
        return a;
    }
    __device__ __forceinline__ s16x8 tr(const char* tile, int st, int db) const {
        const int imm = 16 * D * 2 * st + 512 * db;
        const s16x4 lo = lds_read_tr(tile + tbase[0] + imm);
        const s16x4 hi = lds_read_tr(tile + tbase[1] + imm);
        s16x8 a;
        a[0] = lo[0]; a[1] = lo[1]; a[2] = lo[2]; a[3] = lo[3];
        a[4] = hi[0]; a[5] = hi[1]; a[6] = hi[2]; a[7] = hi[3];
        return a;
    }
};

// Waits until at most N LDS requests younger than the one that produced `a` are outstanding; tied to
// `a` so that the compiler keeps the fragment's first use behind it.
template <int N, typename V>
__device__ __forceinline__ void frag_wait([[maybe_unused]] V& a) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(a) : "n"(N));
#endif
}
template <int N>
__device__ __forceinline__ void fetch_wait() {      // at most N LDS-DMA pieces of this wave still in flight
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory");
#endif
}
// Behind the last of a run of asm-form MFMAs (Mma::acc_v / acc_a), in front of the first vector
// instruction that touches their result: the result latency of 8 passes, which the compiler does not
// know about.
// Tied to the accumulators so that the compiler keeps their readers behind it.
template <typename V>
__device__ __forceinline__ void tie_v([[maybe_unused]] V& c) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+v"(c));
#endif
}
template <typename... V>
__device__ __forceinline__ void mfma_drain_v(V&... c) {
    asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");
    (tie_v(c), ...);
}
template <typename V, int NACC>
__device__ __forceinline__ void mfma_drain_a([[maybe_unused]] V (&c)[NACC]) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");
#pragma unroll
    for (int i = 0; i < NACC; ++i) asm volatile("" : "+a"(c[i]));
#endif
}
__device__ __forceinline__ unsigned lds_addr(const void* p) { return (unsigned)(uintptr_t)(lds_void*)p; }

// Per-lane source offsets of a wave's 1-KiB pieces of a 32 x D tile: piece i of wave w is image bytes
// 1024 (w LPT + i) .. + 1023, lane L at + 16 L, so the image is produced by choosing each lane's
// SOURCE chunk (LDS-DMA writes lane L at base + 16 L).
template <int D, int LPT>
__device__ __forceinline__ void piece_offsets(int wave, int lane, int (&goff)[LPT]) {
#pragma unroll
    for (int i = 0; i < LPT; ++i) {
        const int u = 64 * (wave * LPT + i) + lane;
        const int grp8 = u / D, within = u % D;          // 16-byte units: D of them per 8-row group
        const int row = 8 * grp8 + ((within & 31) >> 2);
        const int ch = 4 * (within >> 5) + ((within & 3) ^ ((row >> 2) & 3));
        goff[i] = (row * D + ch * 8) * 2;
    }
}

// ---------------------------------------------------------------- dQ
// Workgroup = 128 query rows (4 waves x 32), streams K / V tiles of 32 keys.  S^T = K Q^T and
// dP^T = V dO^T put the query on the lane (lse and delta are per-lane scalars); dS^T, converted in
// place, is the B operand of dQ^T += K^T dS^T with K^T from the transposing LDS read.
template <typename T, int D, bool CAUSAL>
__global__ void __launch_bounds__(kThreadsW, 1)
attn_bwd_dq_wide_kernel(const T* __restrict__ Qg, const T* __restrict__ Kg, const T* __restrict__ Vg,
                        const T* __restrict__ dOg, const float* __restrict__ DELg, long long rows,
                        T* __restrict__ dQg, int N, int nqblk, int nwg, float scale) {
    using M = Mma<T>;
    using vec8 = typename M::vec8;
    using vec4 = typename M::vec4;
    constexpr int BQ = 128, TR = kRowsW;
    constexpr int TILE_B = TR * D * 2;
    constexpr int KS = D / 16, DB = D / 32;
    constexpr int LPT = TR * (D / 8) / kThreadsW;
    static_assert(LPT * kThreadsW == TR * (D / 8), "whole 1-KiB pieces per wave");

    constexpr int NBUF = 4;                        // tiles t .. t + 3: two tiles' fetches in flight behind the one in use
    __shared__ __attribute__((aligned(16))) char lds[2 * NBUF * TILE_B];   // K0..K3 V0..V3

    int lid;
    {
        const int bid = blockIdx.x, xcd = bid & 7, idx = bid >> 3, qq = nwg >> 3, rr = nwg & 7;
        lid = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + idx;
    }
    const int head = lid / nqblk;
    const int qblk = CAUSAL ? nqblk - 1 - (lid - head * nqblk) : lid - head * nqblk;   // causal: long blocks first
    const int q0 = qblk * BQ;

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const size_t hoff = (size_t)head * (size_t)N * D;

    const int qrow = q0 + wave * 32 + r;
    const int qrow_c = qrow < N ? qrow : N - 1;
    vec8 qf[KS], dof[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        qf[ks] = *reinterpret_cast<const vec8*>(Qg + hoff + (size_t)qrow_c * D + 16 * ks + 8 * h);
        dof[ks] = *reinterpret_cast<const vec8*>(dOg + hoff + (size_t)qrow_c * D + 16 * ks + 8 * h);
    }
    const float ndelta = DELg[(size_t)head * N + qrow_c];               // -delta
    const float lse2 = DELg[rows + (size_t)head * N + qrow_c];          // lse * log2(e)
    const float sl2 = scale * kLog2eW;

    f32x16 dq[DB];
#pragma unroll
    for (int db = 0; db < DB; ++db)
#pragma unroll
        for (int i = 0; i < 16; ++i) dq[db][i] = 0.f;

    // K and V tiles go HBM/L2 -> LDS directly; rows past N read as zero through the bounds check.
    const unsigned head_bytes = (unsigned)N * D * 2;
    [[maybe_unused]] const __amdgpu_buffer_rsrc_t krs = __builtin_amdgcn_make_buffer_rsrc((void*)(Kg + hoff), 0, head_bytes, 0x00020000);
    [[maybe_unused]] const __amdgpu_buffer_rsrc_t vrs = __builtin_amdgcn_make_buffer_rsrc((void*)(Vg + hoff), 0, head_bytes, 0x00020000);
    int goff[LPT];
    piece_offsets<D, LPT>(wave, lane, goff);
    // piece p of a tile's fetch (p < 2 LPT): even p this wave's KiB piece p/2 of K, odd p that of V.
    // (The host pass of hipcc silently drops a kernel whose body names this builtin.)
    auto fetch_piece = [&](int tile, int buf, int p) __attribute__((always_inline)) {
#if defined(__HIP_DEVICE_COMPILE__)
        if ((p & 1) == 0)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(krs, (lds_void*)(lds + buf * TILE_B + 1024 * (wave * LPT + p / 2)),
                                                     16, goff[p / 2], tile * TILE_B, 0, 0);
        else
            __builtin_amdgcn_raw_ptr_buffer_load_lds(vrs, (lds_void*)(lds + (NBUF + buf) * TILE_B + 1024 * (wave * LPT + p / 2)),
                                                     16, goff[p / 2], tile * TILE_B, 0, 0);
#endif
    };

    const int T_all = (N + TR - 1) / TR;
    const int ntiles = CAUSAL ? min(T_all, (q0 + BQ) / TR) : T_all;
    // tiles 0 and 1 complete, tile 2 in flight (a tile past the end reads zeros without touching memory)
#pragma unroll
    for (int p = 0; p < 2 * LPT; ++p) fetch_piece(0, 0, p);
#pragma unroll
    for (int p = 0; p < 2 * LPT; ++p) fetch_piece(1 < ntiles ? 1 : T_all, 1, p);
#pragma unroll
    for (int p = 0; p < 2 * LPT; ++p) fetch_piece(2 < ntiles ? 2 : T_all, 2, p);
    fetch_wait<2 * LPT>();
    __syncthreads();

    const FragBase<D> fb(lane);
    const unsigned lds0 = lds_addr(lds);
    constexpr int FW = 4;                          // LDS fragments requested ahead of the MFMA stream
    // the fetch of tile t + 3 is spread over the S / dP stream of tile t
    constexpr int FSTEP = 2 * KS / (2 * LPT);
    static_assert(FSTEP >= 2 && 2 * 2 * LPT < 64, "a piece every FSTEP steps; the counted wait fits the counter");

    for (int t = 0; t < ntiles; ++t) {
        const int kv0 = t * TR;
        const int cur = t & (NBUF - 1), fre = (t + 3) & (NBUF - 1);

        f32x16 s, dp;
        {
            const unsigned Ka = lds0 + cur * TILE_B, Va = lds0 + (NBUF + cur) * TILE_B;
            s16x8 win[FW];
            static_for<FW>([&](auto ic) __attribute__((always_inline)) {
                constexpr int i = decltype(ic)::value;
                win[i] = fb.template row_a<i % KS>(i / KS ? Va : Ka);
            });
            static_for<2 * KS>([&](auto ic) __attribute__((always_inline)) {
                constexpr int i = decltype(ic)::value;
                s16x8 a = win[i % FW];
                frag_wait<(2 * KS - 1 - i < FW - 1 ? 2 * KS - 1 - i : FW - 1)>(a);
                // (VGPR-form MFMAs: vector instructions consume S and dP, see Mma::acc_v)
                if constexpr (i == 0) M::first_v(a, qf[0], s);
                else if constexpr (i < KS) M::acc_v(a, qf[i], s);
                else M::acc_v(a, dof[i - KS], dp);
                if constexpr (i + FW < 2 * KS) win[i % FW] = fb.template row_a<(i + FW) % KS>((i + FW) / KS ? Va : Ka);
                // dP^T starts from -delta (two registers a step, under the S stream), so that
                // dS = P o (dP - delta) is one multiplication
                if constexpr (i < 8) { dp[2 * i] = ndelta; dp[2 * i + 1] = ndelta; }
                // P needs S only: its scale-subtract and exp2 run, two elements a step, in the shadow of
                // the MFMAs of dP^T (from the third on: S's last MFMA has to land first)
                if constexpr (i >= KS + 2) {
#pragma unroll
                    for (int e = 2 * (i - KS - 2); e < 2 * (i - KS - 2) + 2; ++e)
                        if (e < 16) s[e] = __builtin_amdgcn_exp2f(__builtin_fmaf(s[e], sl2, -lse2));
                }
                // tile t + 3 into the buffer tile t - 1 was read from (everybody left it at the last
                // barrier; a tile past the end reads zeros): one piece every FSTEP steps (a piece costs
                // the wave some 40 cycles of issue: four in a row under one MFMA stall the stream)
                if constexpr (i % FSTEP == 1 && i / FSTEP < 2 * LPT) fetch_piece(t + 3 < ntiles ? t + 3 : T_all, fre, i / FSTEP);
                __builtin_amdgcn_sched_barrier(0);
            });
        }
        static_assert(2 * (KS - 2) >= 16, "every element of P is formed under the dP stream");
        // accumulator register i is key kv0 + (i & 3) + 8 (i >> 2) + 4 h of the lane's query
        const bool edge = kv0 + TR > N || (CAUSAL && kv0 + TR - 1 > q0 + 32 * wave);
        mfma_drain_v(s, dp);                       // dP's last MFMA has just been issued
#pragma unroll
        for (int i = 0; i < 16; ++i) s[i] *= dp[i];
        if (edge) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int key = kv0 + (i & 3) + 8 * (i >> 2) + 4 * h;
                if (key >= N || (CAUSAL && key > qrow)) s[i] = 0.f;
            }
        }
        vec8 sb[2];
#pragma unroll
        for (int st = 0; st < 2; ++st)
#pragma unroll
            for (int j = 0; j < 8; ++j) sb[st][j] = (T)s[8 * st + j];
        {
            const unsigned Ka = lds0 + cur * TILE_B;
            s16x8 win[FW];
            static_for<FW>([&](auto ic) __attribute__((always_inline)) {
                constexpr int i = decltype(ic)::value;
                win[i] = fb.template tr_a<i / DB, i % DB>(Ka);
            });
            static_for<2 * DB>([&](auto ic) __attribute__((always_inline)) {
                constexpr int i = decltype(ic)::value;
                s16x8 a = win[i % FW];
                frag_wait<2 * (2 * DB - 1 - i < FW - 1 ? 2 * DB - 1 - i : FW - 1)>(a);     // (two reads a fragment)
                M::acc_a(a, sb[i / DB], dq[i % DB]);       // (pinned to accumulator registers: only MFMAs touch dQ in the loop)
                if constexpr (i + FW < 2 * DB) win[i % FW] = fb.template tr_a<(i + FW) / DB, (i + FW) % DB>(Ka);
                __builtin_amdgcn_sched_barrier(0);
            });
        }
        // this wave's pieces of tile t + 1, issued two tiles ago, are in; those of tiles t + 2 and t + 3
        // stay in flight.  Behind the barrier tile t + 1 is complete (all waves' pieces) and tile t
        // released; the bare barrier, not __syncthreads(), which would first wait for the loads in flight
        fetch_wait<2 * 2 * LPT>();
        asm volatile("" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    }

    mfma_drain_a(dq);
    if (qrow < N) {
        T* row = dQg + hoff + (size_t)qrow * D;
#pragma unroll
        for (int db = 0; db < DB; ++db)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                vec4 w;
#pragma unroll
                for (int j = 0; j < 4; ++j) w[j] = (T)(dq[db][4 * g + j] * scale);
                *reinterpret_cast<vec4*>(row + 32 * db + 8 * g + 4 * h) = w;
            }
    }
}

// ---------------------------------------------------------------- dK or dV
// Workgroup = 128 keys (4 waves x 32), streams tiles of 32 queries of Q and dO with the rows'
// lse * log2(e) and -delta.  S = Q K^T (and dP = dO V^T) put the KEY on the lane; P (dS),
// converted in place, is the B operand of dV^T += dO^T P (dK^T += Q^T dS).
//   DV = true:  out = dV = P^T dO            (K fragments only)
//   DV = false: out = dK = scale dS^T Q      (K and V fragments)
template <typename T, int D, bool CAUSAL, bool DV>
__global__ void __launch_bounds__(kThreadsW, 1)
attn_bwd_dkv_wide_kernel(const T* __restrict__ Qg, const T* __restrict__ Kg, const T* __restrict__ Vg,
                         const T* __restrict__ dOg, const float* __restrict__ DELg, long long rows,
                         T* __restrict__ OUTg, int N, int nkblk, int nwg, float scale) {
    using M = Mma<T>;
    using vec8 = typename M::vec8;
    using vec4 = typename M::vec4;
    constexpr int BK = 128, TR = kRowsW;
    constexpr int TILE_B = TR * D * 2;
    constexpr int KS = D / 16, DB = D / 32;
    constexpr int LPT = TR * (D / 8) / kThreadsW;
    static_assert(LPT * kThreadsW == TR * (D / 8), "whole 1-KiB pieces per wave");

    // Q0..Q3 dO0..dO3 (tiles j .. j + 3: two tiles' fetches in flight behind the one in use), then per
    // buffer 64 floats of lse2 and 64 of -delta (a wave-wide 4-byte fetch writes 64 floats: the tile's
    // 32 rows and the 32 behind them, which nobody reads)
    constexpr int NBUF = 4;
    __shared__ __attribute__((aligned(16))) char lds[2 * NBUF * TILE_B + NBUF * 2 * 64 * 4];
    [[maybe_unused]] float* stat = reinterpret_cast<float*>(lds + 2 * NBUF * TILE_B);

    int lid;
    {
        const int bid = blockIdx.x, xcd = bid & 7, idx = bid >> 3, qq = nwg >> 3, rr = nwg & 7;
        lid = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + idx;
    }
    const int head = lid / nkblk;
    const int k0 = (lid - head * nkblk) * BK;      // causal: low key blocks see the most queries and come first

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const size_t hoff = (size_t)head * (size_t)N * D;

    const int krow = k0 + wave * 32 + r;
    const int krow_c = krow < N ? krow : N - 1;
    vec8 kf[KS];
    [[maybe_unused]] vec8 vf[DV ? 1 : KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        kf[ks] = *reinterpret_cast<const vec8*>(Kg + hoff + (size_t)krow_c * D + 16 * ks + 8 * h);
        if constexpr (!DV) vf[ks] = *reinterpret_cast<const vec8*>(Vg + hoff + (size_t)krow_c * D + 16 * ks + 8 * h);
    }
    const float sl2 = scale * kLog2eW;

    const unsigned head_bytes = (unsigned)N * D * 2;
    [[maybe_unused]] const __amdgpu_buffer_rsrc_t qrs = __builtin_amdgcn_make_buffer_rsrc((void*)(Qg + hoff), 0, head_bytes, 0x00020000);
    [[maybe_unused]] const __amdgpu_buffer_rsrc_t ors = __builtin_amdgcn_make_buffer_rsrc((void*)(dOg + hoff), 0, head_bytes, 0x00020000);
    [[maybe_unused]] const __amdgpu_buffer_rsrc_t drs = __builtin_amdgcn_make_buffer_rsrc((void*)(DELg + (size_t)head * N), 0, (unsigned)N * 4, 0x00020000);
    [[maybe_unused]] const __amdgpu_buffer_rsrc_t lrs = __builtin_amdgcn_make_buffer_rsrc((void*)(DELg + rows + (size_t)head * N), 0, (unsigned)N * 4, 0x00020000);
    int goff[LPT];
    piece_offsets<D, LPT>(wave, lane, goff);
    // piece p of a tile's fetch: even p < 2 LPT this wave's KiB piece p/2 of Q, odd p that of dO;
    // piece 2 LPT the row statistics -- by every wave (the same 256 bytes to the same place), so that
    // all waves have the same number of fetches in flight and one counted wait serves them
    constexpr int NPIECE = 2 * LPT + (DV ? 1 : 2);      // fetch instructions per wave and tile
    auto fetch_piece = [&](int tile, int buf, int p) __attribute__((always_inline)) {
#if defined(__HIP_DEVICE_COMPILE__)
        if (p < 2 * LPT && (p & 1) == 0) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(qrs, (lds_void*)(lds + buf * TILE_B + 1024 * (wave * LPT + p / 2)),
                                                     16, goff[p / 2], tile * TILE_B, 0, 0);
        } else if (p < 2 * LPT) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(ors, (lds_void*)(lds + (NBUF + buf) * TILE_B + 1024 * (wave * LPT + p / 2)),
                                                     16, goff[p / 2], tile * TILE_B, 0, 0);
        } else {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(lrs, (lds_void*)(stat + buf * 128), 4, lane * 4, tile * TR * 4, 0, 0);
            if constexpr (!DV)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(drs, (lds_void*)(stat + buf * 128 + 64), 4, lane * 4, tile * TR * 4, 0, 0);
        }
#endif
    };

    f32x16 acc[DB];
#pragma unroll
    for (int db = 0; db < DB; ++db)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[db][i] = 0.f;

    const int T_all = (N + TR - 1) / TR;
    const int t0 = CAUSAL ? k0 / TR : 0;           // causal: queries before the block's first key see none of it
    // tiles t0 and t0 + 1 complete, tile t0 + 2 in flight (a tile past the end reads zeros)
#pragma unroll
    for (int p = 0; p <= 2 * LPT; ++p) fetch_piece(t0, 0, p);
#pragma unroll
    for (int p = 0; p <= 2 * LPT; ++p) fetch_piece(t0 + 1 < T_all ? t0 + 1 : T_all, 1, p);
#pragma unroll
    for (int p = 0; p <= 2 * LPT; ++p) fetch_piece(t0 + 2 < T_all ? t0 + 2 : T_all, 2, p);
    fetch_wait<NPIECE>();
    __syncthreads();

    const FragBase<D> fb(lane);
    const unsigned lds0 = lds_addr(lds);
    constexpr int FW = 4;
    constexpr int NA = DV ? KS : 2 * KS;           // MFMAs of the tile's first stream
    static_assert((NA + 2 * DB - 2) / 3 >= 2 * LPT && 2 * NPIECE < 64, "every piece has its step; the counted wait fits the counter");

    for (int j = t0; j < T_all; ++j) {
        const int qt0 = j * TR;
        const int cur = (j - t0) & (NBUF - 1), fre = (j - t0 + 3) & (NBUF - 1);
        const unsigned Qa = lds0 + cur * TILE_B, dOa = lds0 + (NBUF + cur) * TILE_B;

        // accumulator register i is query qt0 + (i & 3) + 8 (i >> 2) + 4 h: the rows' lse2 come as four
        // broadcast float4 reads, and four more start the dP accumulators at -delta.  (Asm reads like the
        // fragments': a read the compiler knows as one would wait for the fetches in flight.)
        f32x16 s, dp;
        f32x4 l4[4];
        [[maybe_unused]] f32x4 nd4[4];
        {
            [[maybe_unused]] const unsigned sa = lds0 + 2 * NBUF * TILE_B + cur * 512 + 16 * h;
#if defined(__HIP_DEVICE_COMPILE__)
            static_for<4>([&](auto gc) __attribute__((always_inline)) {
                constexpr int g = decltype(gc)::value;
                [[maybe_unused]] auto& nd = nd4;           // (captured here: inside the discarded branch clang does not)
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(l4[g]) : "v"(sa), "n"(32 * g));
                if constexpr (!DV) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(nd[g]) : "v"(sa), "n"(256 + 32 * g));
            });
#endif
        }
        {
            s16x8 win[FW];
            static_for<FW>([&](auto ic) __attribute__((always_inline)) {
                constexpr int i = decltype(ic)::value;
                win[i] = fb.template row_a<i % KS>(i / KS ? dOa : Qa);
            });
            // the statistics are older than the FW fragment requests behind them
#if defined(__HIP_DEVICE_COMPILE__)
            if constexpr (DV) asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(l4[0]), "+v"(l4[1]), "+v"(l4[2]), "+v"(l4[3]) : "n"(FW));
            else asm volatile("s_waitcnt lgkmcnt(%8)" : "+v"(l4[0]), "+v"(l4[1]), "+v"(l4[2]), "+v"(l4[3]),
                              "+v"(nd4[0]), "+v"(nd4[1]), "+v"(nd4[2]), "+v"(nd4[3]) : "n"(FW));
#endif
            if constexpr (!DV) {
#pragma unroll
                for (int g = 0; g < 4; ++g) { dp[4 * g] = nd4[g][0]; dp[4 * g + 1] = nd4[g][1]; dp[4 * g + 2] = nd4[g][2]; dp[4 * g + 3] = nd4[g][3]; }
            }
            static_for<NA>([&](auto ic) __attribute__((always_inline)) {
                constexpr int i = decltype(ic)::value;
                s16x8 a = win[i % FW];
                frag_wait<(NA - 1 - i < FW - 1 ? NA - 1 - i : FW - 1)>(a);
                // (VGPR-form MFMAs: vector instructions consume S and dP, see Mma::acc_v)
                if constexpr (i == 0) M::first_v(a, kf[0], s);
                else if constexpr (i < KS) M::acc_v(a, kf[i], s);
                else if constexpr (!DV) M::acc_v(a, vf[i - KS], dp);
                if constexpr (i + FW < NA) win[i % FW] = fb.template row_a<(i + FW) % KS>((i + FW) / KS ? dOa : Qa);
                // tile j + 3 into the buffer tile j - 1 was read from: one piece every third step of the
                // tile's two streams (a piece costs the wave some 40 cycles of issue)
                if constexpr (i % 3 == 1 && i / 3 <= 2 * LPT) fetch_piece(j + 3 < T_all ? j + 3 : T_all, fre, i / 3);
                // dK pass: P needs S only, so it is formed in the shadow of the MFMAs of dP (from the
                // third on: S's last MFMA has to land first), two elements a step
                if constexpr (!DV && i >= KS + 2) {
#pragma unroll
                    for (int e = 2 * (i - KS - 2); e < 2 * (i - KS - 2) + 2; ++e)
                        if (e < 16) s[e] = __builtin_amdgcn_exp2f(__builtin_fmaf(s[e], sl2, -l4[e >> 2][e & 3]));
                }
                __builtin_amdgcn_sched_barrier(0);
            });
        }
        if constexpr (DV) mfma_drain_v(s);         // the stream's last MFMA has just been issued
        else mfma_drain_v(s, dp);
        static_assert(2 * (KS - 2) >= 16, "every element of P is formed under the dP stream");
        // pairs (query, key) outside the problem or above the diagonal: element c = 8 (i >> 2) + (i & 3)
        // of this lane is masked iff c >= c_hi or c < c_lo
        const bool edge = qt0 + TR > N || k0 + BK > N || (CAUSAL && qt0 < k0 + BK);
        const int c_hi = krow < N ? N - (qt0 + 4 * h) : -1;
        const int c_lo = CAUSAL ? krow - (qt0 + 4 * h) : -(1 << 30);
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if constexpr (DV) s[i] = __builtin_amdgcn_exp2f(__builtin_fmaf(s[i], sl2, -l4[i >> 2][i & 3]));
            else s[i] *= dp[i];
        }
        if (edge) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int c = 8 * (i >> 2) + (i & 3);
                if (c >= c_hi || c < c_lo) s[i] = 0.f;
            }
        }
        vec8 pb[2];
#pragma unroll
        for (int st = 0; st < 2; ++st)
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) pb[st][jj] = (T)s[8 * st + jj];
        {
            const unsigned tile = DV ? dOa : Qa;
            s16x8 win[FW];
            static_for<FW>([&](auto ic) __attribute__((always_inline)) {
                constexpr int i = decltype(ic)::value;
                win[i] = fb.template tr_a<i / DB, i % DB>(tile);
            });
            static_for<2 * DB>([&](auto ic) __attribute__((always_inline)) {
                constexpr int i = decltype(ic)::value;
                s16x8 a = win[i % FW];
                frag_wait<2 * (2 * DB - 1 - i < FW - 1 ? 2 * DB - 1 - i : FW - 1)>(a);     // (two reads a fragment)
                M::acc_a(a, pb[i / DB], acc[i % DB]);      // (pinned to accumulator registers: only MFMAs touch it in the loop)
                if constexpr (i + FW < 2 * DB) win[i % FW] = fb.template tr_a<(i + FW) / DB, (i + FW) % DB>(tile);
                if constexpr ((NA + i) % 3 == 1 && (NA + i) / 3 <= 2 * LPT) fetch_piece(j + 3 < T_all ? j + 3 : T_all, fre, (NA + i) / 3);
                __builtin_amdgcn_sched_barrier(0);
            });
        }
        // this wave's pieces of tile j + 1, issued two tiles ago, are in; those of tiles j + 2 and j + 3
        // stay in flight
        fetch_wait<2 * NPIECE>();
        asm volatile("" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    }

    mfma_drain_a(acc);
    if (krow < N) {
        T* row = OUTg + hoff + (size_t)krow * D;
        const float f = DV ? 1.0f : scale;
#pragma unroll
        for (int db = 0; db < DB; ++db)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                vec4 w;
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) w[jj] = (T)(acc[db][4 * g + jj] * f);
                *reinterpret_cast<vec4*>(row + 32 * db + 8 * g + 4 * h) = w;
            }
    }
}

template <typename T, int D, bool CAUSAL>
void launch_three(const T* Q, const T* K, const T* V, const T* dO, const float* delta, int64_t rows,
                  T* dq, T* dk, T* dv, int64_t N, int64_t nqblk, int64_t nq, int64_t nkblk, int64_t nk,
                  float scale, hipStream_t s) {
    launch_kernel((attn_bwd_dkv_wide_kernel<T, D, CAUSAL, true>), dim3((unsigned)nk), dim3(kThreadsW), 0, s, Q, K, V, dO, delta,
                  (long long)rows, dv, (int)N, (int)nkblk, (int)nk, scale);
    launch_kernel((attn_bwd_dkv_wide_kernel<T, D, CAUSAL, false>), dim3((unsigned)nk), dim3(kThreadsW), 0, s, Q, K, V, dO, delta,
                  (long long)rows, dk, (int)N, (int)nkblk, (int)nk, scale);
    launch_kernel((attn_bwd_dq_wide_kernel<T, D, CAUSAL>), dim3((unsigned)nq), dim3(kThreadsW), 0, s, Q, K, V, dO, delta,
                  (long long)rows, dq, (int)N, (int)nqblk, (int)nq, scale);
}

template <typename T, int D>
int launch_bwd_wide(const void* q, const void* k, const void* v, const void* o, const float* lse, const void* dout,
                    void* dq, void* dk, void* dv, float* delta, int64_t heads, int64_t N, float scale,
                    bool causal, hipStream_t s) {
    const int64_t rows = heads * N;
    // every size check comes before the first launch
    const int64_t nqblk = cdiv(N, 128), nkblk = cdiv(N, 128);
    const int64_t nq = heads * nqblk, nk = heads * nkblk;
    if (nq > 0x7fffffffLL || cdiv(rows, 16) > 0x7fffffffLL)
        return fail(MYFA_E_UNSUPPORTED, "attention backward: too many workgroups");
    // byte offsets inside a head are 32-bit, up to two tiles past its end
    if ((N + 4 * kRowsW) * D * 2 > 0x7fffffffLL)
        return fail(MYFA_E_UNSUPPORTED, "attention backward: N=%lld too large for head dim %d", (long long)N, D);
    const T* Q = static_cast<const T*>(q);
    const T* K = static_cast<const T*>(k);
    const T* V = static_cast<const T*>(v);
    const T* dO = static_cast<const T*>(dout);
    launch_kernel((attn_bwd_wide_delta_kernel<T, D>), dim3((unsigned)cdiv(rows, 16)), dim3(256), 0, s,
                  static_cast<const T*>(o), dO, lse, delta, (long long)rows);
    if (causal)
        launch_three<T, D, true>(Q, K, V, dO, delta, rows, static_cast<T*>(dq), static_cast<T*>(dk), static_cast<T*>(dv),
                                 N, nqblk, nq, nkblk, nk, scale, s);
    else
        launch_three<T, D, false>(Q, K, V, dO, delta, rows, static_cast<T*>(dq), static_cast<T*>(dk), static_cast<T*>(dv),
                                  N, nqblk, nq, nkblk, nk, scale, s);
    return check_launch("attn_bwd_wide");
}

}  // namespace

// dtype: MYFA_F16 or MYFA_BF16; d: 192 or 256; heads = B*H.  `delta`: float32 workspace of
// 2 * heads * N elements.  Called by attn_bwd_mfma() (attn_bwd_mfma.hip) for these head dims.
int attn_bwd_wide(const void* q, const void* k, const void* v, const void* o, const float* lse, const void* dout,
                  void* dq, void* dk, void* dv, float* delta, int64_t heads, int64_t N, int64_t d,
                  float scale, int dtype, bool causal, hipStream_t s) {
    if (dtype == MYFA_BF16 && d == 256) return launch_bwd_wide<__bf16, 256>(q, k, v, o, lse, dout, dq, dk, dv, delta, heads, N, scale, causal, s);
    if (dtype == MYFA_F16 && d == 256)  return launch_bwd_wide<_Float16, 256>(q, k, v, o, lse, dout, dq, dk, dv, delta, heads, N, scale, causal, s);
    if (dtype == MYFA_BF16 && d == 192) return launch_bwd_wide<__bf16, 192>(q, k, v, o, lse, dout, dq, dk, dv, delta, heads, N, scale, causal, s);
    if (dtype == MYFA_F16 && d == 192)  return launch_bwd_wide<_Float16, 192>(q, k, v, o, lse, dout, dq, dk, dv, delta, heads, N, scale, causal, s);
    return fail(MYFA_E_UNSUPPORTED, "attention backward (wide): needs float16 or bfloat16 and head dim 192 or 256 (got dtype %d, d %lld)", dtype, (long long)d);
}

}  // namespace myfa
