// Attention backward for f16 / bf16 on gfx950 matrix cores.
//
//     S = scale Q K^T,  P = exp(S - lse),  O = P V           (forward, lse saved per row)
//     dV = P^T dO
//     dP = dO V^T,  delta = rowsum(dO o O),  dS = P o (dP - delta)
//     dQ = scale dS K,   dK = scale dS^T Q
//
// The reference has no backward at all (its forward kernel was never written,
// csrc/my_flash_attn/flash_attention.cu:23-69, and setup.py:128 only hints at upstream's
// DISABLE_BACKWARD switch); this is the natural continuation of the fused forward, in the
// formulation of FlashAttention-2's backward with the row statistic m + log l saved by the
// forward as one array (the l, m of tests/test_flash_attn.py:58-59,98-99).
//
// Three kernels, no atomics, deterministic:
//   delta   one pass over dO and O.
//   dQ      workgroup = 256 query rows (8 waves x 32), streams K/V tiles of 64 keys -- the
//           forward kernel's arrangement: S^T = K Q^T and dP^T = V dO^T put the query on
//           the lane, P and dS are formed per lane with the row's lse and delta as scalars,
//           and dS^T (converted in place) is the B operand of dQ^T += K^T dS^T, K^T coming
//           from the row-major K tile through the transposing LDS read.
//   dK,dV   the mirror image: workgroup = 128 keys (4 waves x 32, one wave per SIMD so that
//           K, V fragments and both accumulators fit its 512 registers), streams tiles of 64
//           queries of Q and dO.  S = Q K^T and dP = dO V^T put the KEY on the lane;
//           lse and delta vary along the accumulator's rows and come from LDS;
//           dV^T += dO^T P and dK^T += Q^T dS take P and dS as B operands in place.
// P is recomputed in each of the two kernels (7 tile products in all instead of 5 with
// atomics on dQ).
#include "attn_mfma_common.h"

#include <mutex>

namespace myfa {
namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr float kLog2e = 1.4426950408889634f;

constexpr int kTile = 64;            // rows of a streamed tile (keys in dQ, queries in dK/dV)

// ---------------------------------------------------------------- delta = rowsum(dO o O)
// (stored NEGATED, so that the dK/dV kernel can start its dP accumulators from it, and the row
// statistic in base 2, lse * log2(e), next to it: the two per-row scalars the other kernels
// consume, in one workspace of 2 * rows floats: [-delta | lse2])
template <typename T, int D>
__global__ void __launch_bounds__(256)
attn_bwd_delta_kernel(const T* __restrict__ Og, const T* __restrict__ dOg, const float* __restrict__ LSEg,
                      float* __restrict__ DELg, long long rows) {
    using vec8 = typename Mma<T>::vec8;
    constexpr int CPR = D / 8;                     // 16-byte chunks per row = threads per row
    const long long row = (long long)blockIdx.x * (256 / CPR) + threadIdx.x / CPR;
    const int ch = threadIdx.x % CPR;
    float acc = 0.f;
    if (row < rows) {
        const vec8 a = *reinterpret_cast<const vec8*>(Og + row * D + ch * 8);
        const vec8 b = *reinterpret_cast<const vec8*>(dOg + row * D + ch * 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) acc += (float)a[j] * (float)b[j];
    }
    // sum over the CPR (8 or 16) adjacent lanes of a row, on DPP: xor 1, xor 2 inside quads, then the
    // mirrored quad of the 8-lane group, then the mirrored half of the 16-lane row
    acc += dpp_mov<0xB1>(acc);
    acc += dpp_mov<0x4E>(acc);
    acc += dpp_mov<0x141>(acc);
    if constexpr (CPR == 16) acc += dpp_mov<0x140>(acc);
    static_assert(CPR == 8 || CPR == 16, "row sum written for 8 or 16 lanes per row");
    if (row < rows && ch == 0) {
        DELg[row] = -acc;
        DELg[rows + row] = LSEg[row] * kLog2e;
    }
}

// ---------------------------------------------------------------- dQ
template <typename T, int D, bool CAUSAL>
__global__ void __launch_bounds__(512, 2)
attn_bwd_dq_kernel(const T* __restrict__ Qg, const T* __restrict__ Kg, const T* __restrict__ Vg,
                   const T* __restrict__ dOg, const float* __restrict__ LSEg, const float* __restrict__ DELg,
                   T* __restrict__ dQg, int N, int nqblk, int nwg, float scale) {
    using M = Mma<T>;
    using vec8 = typename M::vec8;
    using vec4 = typename M::vec4;
    constexpr int THREADS = 512, BQ = 256;
    constexpr int TILE_B = kTile * D * 2;
    constexpr int KS = D / 16, DB = D / 32;

    // K0 K1 K2 V0 V1 V2: three buffers, the loads of tile t + 2 go out in the last steps of tile t
    // (see the dK/dV kernel for why there and not at the top of the tile)
    __shared__ __attribute__((aligned(16))) char lds[6 * TILE_B];

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
    const T* Kh = Kg + hoff;
    const T* Vh = Vg + hoff;

    const int qrow = q0 + wave * 32 + r;
    const int qrow_c = qrow < N ? qrow : N - 1;
    vec8 qf[KS], dof[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        qf[ks] = *reinterpret_cast<const vec8*>(Qg + hoff + (size_t)qrow_c * D + 16 * ks + 8 * h);
        dof[ks] = *reinterpret_cast<const vec8*>(dOg + hoff + (size_t)qrow_c * D + 16 * ks + 8 * h);
    }
    const float lse2 = LSEg[(size_t)head * N + qrow_c] * kLog2e;
    const float ndelta = DELg[(size_t)head * N + qrow_c];      // -delta
    const float sl2 = scale * kLog2e;

    f32x16 dq[DB];
#pragma unroll
    for (int db = 0; db < DB; ++db)
#pragma unroll
        for (int i = 0; i < 16; ++i) dq[db][i] = 0.f;

    // K and V tiles go HBM/L2 -> LDS directly (buffer_load ... lds), the image produced by choosing
    // each lane's source chunk; rows past N read as zero through the buffer bounds check.
    constexpr int CPR = D / 8, LPT = kTile * CPR / THREADS, SUB_B = 16 * D;
    const unsigned head_bytes = (unsigned)N * D * 2;
    [[maybe_unused]] const __amdgpu_buffer_rsrc_t krs = __builtin_amdgcn_make_buffer_rsrc((void*)Kh, 0, head_bytes, 0x00020000);
    [[maybe_unused]] const __amdgpu_buffer_rsrc_t vrs = __builtin_amdgcn_make_buffer_rsrc((void*)Vh, 0, head_bytes, 0x00020000);
    int st_goff[LPT];
#pragma unroll
    for (int i = 0; i < LPT; ++i) {
        const int u = 64 * (wave * LPT + i) + lane;
        const int grp8 = u / (SUB_B / 16), within = u % (SUB_B / 16);
        const int row = 8 * grp8 + ((within & 31) >> 2);
        const int ch = 4 * (within >> 5) + ((within & 3) ^ ((row >> 2) & 3));
        st_goff[i] = (row * D + ch * 8) * 2;
    }
    // (the host pass of hipcc silently drops a kernel whose body names this builtin; the calls
    // only exist in the device pass)
    // piece p of a tile's fetch: even p this wave's KiB piece p/2 of K, odd p that of V (p < 2 LPT)
    auto fetch_piece = [&](int tile, int buf, int p) __attribute__((always_inline)) {
#if defined(__HIP_DEVICE_COMPILE__)
        if ((p & 1) == 0)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(krs, (lds_void*)(lds + buf * TILE_B + 1024 * (wave * LPT + p / 2)),
                                                     16, st_goff[p / 2], tile * TILE_B, 0, 0);
        else
            __builtin_amdgcn_raw_ptr_buffer_load_lds(vrs, (lds_void*)(lds + (3 + buf) * TILE_B + 1024 * (wave * LPT + p / 2)),
                                                     16, st_goff[p / 2], tile * TILE_B, 0, 0);
#endif
    };
    auto fetch_tile = [&](int tile, int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int p = 0; p < 2 * LPT; ++p) fetch_piece(tile, buf, p);
    };

    const int T_all = (N + kTile - 1) / kTile;
    const int ntiles = CAUSAL ? min(T_all, (q0 + BQ) / kTile) : T_all;
    fetch_tile(0, 0);
    fetch_tile(1 < ntiles ? 1 : T_all, 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int cur = 0, nxt = 1, fre = 2;                 // buffers of tile t, t + 1, t + 2

    // Per-lane bases of the fragment reads; everything else in an address is an immediate.
    const int tq = (lane & 15) >> 2, tp = lane & 3, tg1 = (lane >> 4) & 1;
    int rbase[2], tbase[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        rbase[e] = SUB_B * (r >> 3) + 64 * (r & 7) + 16 * ((2 * e + h) ^ ((r >> 2) & 3));
        tbase[e] = 64 * (4 * h + tq) + 16 * ((2 * tg1 + (tp >> 1)) ^ (2 * e + h)) + 8 * (tp & 1) + SUB_B * e;
    }
    // An MFMA sequence as one software-pipelined stream (LDS fragment of MFMA i + FW requested
    // when MFMA i issues, order pinned); `after(i)` runs in the shadow of MFMA i.
    constexpr int FW = 4;
    auto stream = [&](auto n_, auto&& frag, auto&& mma, auto&& after) __attribute__((always_inline)) {
        constexpr int NFR = decltype(n_)::value;
        s16x8 win[FW];
#pragma unroll
        for (int i = 0; i < FW && i < NFR; ++i) win[i] = frag(i);
#pragma unroll
        for (int i = 0; i < NFR; ++i) {
            const s16x8 a = win[i % FW];
            mma(i, a);
            if (i + FW < NFR) win[i % FW] = frag(i + FW);
            after(i);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    auto nothing = [](int) __attribute__((always_inline)) {};

    for (int t = 0; t < ntiles; ++t) {
        const int kv0 = t * kTile;
        const char* Kb = lds + cur * TILE_B;
        const char* Vb = lds + (3 + cur) * TILE_B;

        // The tile's 64 keys are two blocks of 32, staggered so that the vector part of one block
        // runs in the shadow of the other block's MFMAs:
        //   A  S0^T, dP0^T                 B  S1^T, dP1^T   with dS0^T
        //   C  dQ^T += block 0  with dS1^T  D  dQ^T += block 1
        // S^T = K Q^T and dP^T = V dO^T put the query on the lane, so lse2 and delta are scalars.
        const bool edge = kv0 + kTile > N || (CAUSAL && kv0 + kTile - 1 > q0 + 32 * wave);
        f32x16 s[2], dp[2];
        vec8 sb[4];
        auto pds = [&](int kb, int i) __attribute__((always_inline)) {
            const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(s[kb][i], sl2, -lse2));
            s[kb][i] = p * (dp[kb][i] + ndelta);
        };
        auto finish = [&](int kb) __attribute__((always_inline)) {
            if (edge) {
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int key = kv0 + 32 * kb + (i & 3) + 8 * (i >> 2) + 4 * h;
                    if (key >= N || (CAUSAL && key > qrow)) s[kb][i] = 0.f;
                }
            }
#pragma unroll
            for (int st = 0; st < 2; ++st)
#pragma unroll
                for (int j = 0; j < 8; ++j) sb[2 * kb + st][j] = (T)s[kb][8 * st + j];
        };
        auto row_frag = [&](int kb, int i) __attribute__((always_inline)) -> s16x8 {
            const int which = i / KS, ks = i % KS;
            const char* tile = which ? Vb : Kb;
            return *reinterpret_cast<const s16x8*>(tile + rbase[ks & 1] + SUB_B * 4 * kb + 512 * (ks >> 1));
        };
        auto row_mma = [&](int kb, int i, s16x8 a) __attribute__((always_inline)) {
            const int which = i / KS, ks = i % KS;
            if (which) dp[kb] = M::run(a, dof[ks], dp[kb]);
            else s[kb] = M::run(a, qf[ks], s[kb]);
        };
        auto tr_frag = [&](int kb, int i) __attribute__((always_inline)) -> s16x8 {
            const int st = i / DB, db = i % DB;
            const int imm = SUB_B * 2 * (2 * kb + st) + 512 * db;
            const s16x4 lo = lds_read_tr(Kb + tbase[0] + imm);
            const s16x4 hi = lds_read_tr(Kb + tbase[1] + imm);
            s16x8 a;
            a[0] = lo[0]; a[1] = lo[1]; a[2] = lo[2]; a[3] = lo[3];
            a[4] = hi[0]; a[5] = hi[1]; a[6] = hi[2]; a[7] = hi[3];
            return a;
        };
        auto tr_mma = [&](int kb, int i, s16x8 a) __attribute__((always_inline)) {
            const int st = i / DB, db = i % DB;
            dq[db] = M::run(a, sb[2 * kb + st], dq[db]);
        };
        constexpr int EPB = 16 / (2 * KS), EPC = 16 / (2 * DB);   // elements per MFMA step in B and in C

#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int i = 0; i < 16; ++i) { s[kb][i] = 0.f; dp[kb][i] = 0.f; }
        stream(IC<2 * KS>{},                                                                       // A
               [&](int i) __attribute__((always_inline)) -> s16x8 { return row_frag(0, i); },
               [&](int i, s16x8 a) __attribute__((always_inline)) { row_mma(0, i, a); }, nothing);
        stream(IC<2 * KS>{},                                                                       // B
               [&](int i) __attribute__((always_inline)) -> s16x8 { return row_frag(1, i); },
               [&](int i, s16x8 a) __attribute__((always_inline)) { row_mma(1, i, a); },
               [&](int i) __attribute__((always_inline)) {
#pragma unroll
                   for (int e = 0; e < EPB; ++e) pds(0, EPB * i + e);
               });
        finish(0);
        stream(IC<2 * DB>{},                                                                       // C
               [&](int i) __attribute__((always_inline)) -> s16x8 { return tr_frag(0, i); },
               [&](int i, s16x8 a) __attribute__((always_inline)) { tr_mma(0, i, a); },
               [&](int i) __attribute__((always_inline)) {
#pragma unroll
                   for (int e = 0; e < EPC; ++e) pds(1, EPC * i + e);
               });
        finish(1);
        stream(IC<2 * DB>{},                                                                       // D
               [&](int i) __attribute__((always_inline)) -> s16x8 { return tr_frag(1, i); },
               [&](int i, s16x8 a) __attribute__((always_inline)) { tr_mma(1, i, a); },
               [&](int i) __attribute__((always_inline)) {
                   // Tile t + 2 into the buffer tile t - 1 was read from (everybody left it at the last
                   // barrier; unconditional, a tile past the end reads zeros): one piece per step over
                   // the phase's last steps, behind the tile's last fragment request.  The wait in
                   // front confirms this wave's pieces of tile t + 1, issued a tile ago.
                   constexpr int P0 = 2 * DB - 2 * LPT;
                   static_assert(P0 >= 0 && (2 * DB <= FW || P0 >= 2 * DB - FW), "the loads follow the last fragment request");
                   if (i == P0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                   // (under the causal mask the tiles behind the block's last one exist but are never
                   // read: asked for as the tile past the end, the bounds check answers with zeros)
                   if (i >= P0) fetch_piece(t + 2 < ntiles ? t + 2 : T_all, fre, i - P0);
               });

        // tile t + 1 complete (all waves' pieces), tile t released; the bare barrier, not
        // __syncthreads(), which would first wait for the loads issued a moment ago
        asm volatile("" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        const int was = cur;
        cur = nxt; nxt = fre; fre = was;
    }

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

// ---------------------------------------------------------------- dK, dV
// The vector work of one tile of the dK/dV kernel, planned at compile time.
//
// One wave per SIMD means nothing but the wave's own MFMAs can cover its vector instructions, and
// the wave issues ONE instruction at a time: everything that is not an MFMA has to fit, in issue
// slots, between MFMAs that each keep the matrix core busy for 32 cycles.  A tile is 8 KS MFMA
// steps (g = 0 .. 8 KS - 1):
//     A  [0, 2KS)    S0 = Q0 K^T (KS steps), dP0 = dO0 V^T (KS)        block 0 = queries 0..31
//     B  [2KS, 4KS)  S1, dP1                                            block 1 = queries 32..63
//     C  [4KS, 6KS)  dV += dO0^T P0, dK += Q0^T dS0, alternating; its first half takes elements
//                    0..7 of the block's accumulators, its second half elements 8..15
//     D  [6KS, 8KS)  the same for block 1
// and per accumulator element e of a block the vector work is
//     kind 0   t = s * scale*log2e - lse2          (needs S: two steps behind its last MFMA)
//     kind 1   p = exp2(t)
//     kind 2   ds = p * dp                         (needs dP, which started from -delta)
//     kind 3   P pair (2e, 2e+1) -> 16 bit         kind 4   dS pair -> 16 bit
// plan_dkv() deals these 128 operations onto the steps: earliest deadline first, an operation
// never in the same step as the one it depends on, at most `cap` cycles of issue per step.
struct DkvOp { int blk, kind, e; };
template <int KS> struct DkvPlan {
    static constexpr int NSTEP = 8 * KS, NOPS = 128;
    DkvOp op[NOPS] = {};
    int start[NSTEP + 1] = {};
    bool ok = false;
};
template <int KS>
constexpr DkvPlan<KS> plan_dkv(int cap) {
    constexpr int cost[5] = {5, 9, 5, 8, 8};
    DkvPlan<KS> p{};
    int at[2][5][16] = {};
    for (int b = 0; b < 2; ++b)
        for (int k = 0; k < 5; ++k)
            for (int e = 0; e < 16; ++e) at[b][k][e] = -1;
    int n = 0;
    bool late = false;
    for (int g = 0; g < p.NSTEP; ++g) {
        p.start[g] = n;
        int budget = cap;
        while (true) {
            int bb = -1, bk = -1, be = -1, bdl = 1 << 30;
            for (int b = 0; b < 2; ++b) {
                const int s_rdy = (2 * b + 1) * KS + 1, dp_rdy = (2 * b + 2) * KS + 1;
                for (int k = 4; k >= 0; --k)
                    for (int e = 0; e < (k < 3 ? 16 : 8); ++e) {
                        if (at[b][k][e] >= 0) continue;
                        bool rdy = false;
                        if (k == 0) rdy = g >= s_rdy;
                        if (k == 1) rdy = at[b][0][e] >= 0 && at[b][0][e] < g;
                        if (k == 2) rdy = g >= dp_rdy && at[b][1][e] >= 0 && at[b][1][e] < g;
                        if (k == 3) rdy = at[b][1][2 * e] >= 0 && at[b][1][2 * e] < g && at[b][1][2 * e + 1] >= 0 && at[b][1][2 * e + 1] < g;
                        if (k == 4) rdy = at[b][2][2 * e] >= 0 && at[b][2][2 * e] < g && at[b][2][2 * e + 1] >= 0 && at[b][2][2 * e + 1] < g;
                        if (!rdy) continue;
                        const int e0 = k < 3 ? e : 2 * e;
                        const int dl = (4 + 2 * b) * KS + (e0 >= 8 ? KS : 0) - 1;     // last step it may run behind
                        if (dl < bdl) { bdl = dl; bb = b; bk = k; be = e; }           // (ties: later kind, lower element)
                    }
            }
            if (bb < 0 || cost[bk] > budget) break;
            budget -= cost[bk];
            at[bb][bk][be] = g;
            p.op[n++] = DkvOp{bb, bk, be};
            if (g > bdl) late = true;
        }
    }
    p.start[p.NSTEP] = n;
    p.ok = n == p.NOPS && !late;
    return p;
}
// the smallest issue budget per step (in cycles) that meets every deadline
template <int KS, int CAP = 16, bool OK = plan_dkv<KS>(CAP).ok> struct DkvPlanFor;
template <int KS, int CAP> struct DkvPlanFor<KS, CAP, true> {
    static constexpr int cap = CAP;
    static constexpr DkvPlan<KS> plan = plan_dkv<KS>(CAP);
};
template <int KS, int CAP> struct DkvPlanFor<KS, CAP, false> : DkvPlanFor<KS, CAP + 2> {};

template <typename T, int D, bool CAUSAL>
__global__ void __launch_bounds__(256, 1)
attn_bwd_dkv_kernel(const T* __restrict__ Qg, const T* __restrict__ Kg, const T* __restrict__ Vg,
                    const T* __restrict__ dOg, const float* __restrict__ LSE2g, const float* __restrict__ DELg,
                    T* __restrict__ dKg, T* __restrict__ dVg, int N, int nkblk, int nwg, float scale) {
    using M = Mma<T>;
    using vec8 = typename M::vec8;
    using vec4 = typename M::vec4;
    constexpr int THREADS = 256, BK = 128;
    constexpr int TILE_B = kTile * D * 2;
    constexpr int KS = D / 16, DB = D / 32;

    // Q0 Q1 Q2 dO0 dO1 dO2, then lse*log2e and -delta of the tile's 64 queries.  Three buffers: the
    // requests for tile j + 2 are issued in the LAST steps of tile j (see the loop), into the buffer
    // tile j - 1 was read from, while tile j + 1 is already complete.
    __shared__ __attribute__((aligned(16))) char lds[6 * TILE_B + 6 * kTile * 4];
    float* stat = reinterpret_cast<float*>(lds + 6 * TILE_B);     // [buf][lse2 | -delta][64]

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
    const T* Qh = Qg + hoff;
    const T* dOh = dOg + hoff;
    const float* lseh = LSE2g + (size_t)head * N;    // lse * log2(e), written by the delta kernel
    const float* delh = DELg + (size_t)head * N;

    const int krow = k0 + wave * 32 + r;
    const int krow_c = krow < N ? krow : N - 1;
    vec8 kf[KS], vf[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        kf[ks] = *reinterpret_cast<const vec8*>(Kg + hoff + (size_t)krow_c * D + 16 * ks + 8 * h);
        vf[ks] = *reinterpret_cast<const vec8*>(Vg + hoff + (size_t)krow_c * D + 16 * ks + 8 * h);
    }
    const float sl2 = scale * kLog2e;

    // Q and dO tiles (and the 64 rows' lse2 / delta) go HBM/L2 -> LDS directly (buffer_load ... lds):
    // one wave-instruction fills 1 KiB of a tile image, lane L at base + 16 L, so the image is
    // produced by choosing each lane's SOURCE chunk, as in the forward kernel.  Rows past N read
    // as zero through the buffer bounds check (and are masked as edge rows).
    constexpr int CPR = D / 8, LPT = kTile * CPR / THREADS;
    const unsigned head_bytes = (unsigned)N * D * 2;
    [[maybe_unused]] const __amdgpu_buffer_rsrc_t qrs = __builtin_amdgcn_make_buffer_rsrc((void*)Qh, 0, head_bytes, 0x00020000);
    [[maybe_unused]] const __amdgpu_buffer_rsrc_t ors = __builtin_amdgcn_make_buffer_rsrc((void*)dOh, 0, head_bytes, 0x00020000);
    [[maybe_unused]] const __amdgpu_buffer_rsrc_t lrs = __builtin_amdgcn_make_buffer_rsrc((void*)lseh, 0, (unsigned)N * 4, 0x00020000);
    [[maybe_unused]] const __amdgpu_buffer_rsrc_t drs = __builtin_amdgcn_make_buffer_rsrc((void*)delh, 0, (unsigned)N * 4, 0x00020000);
    int st_goff[LPT];
#pragma unroll
    for (int i = 0; i < LPT; ++i) {
        const int u = 64 * (wave * LPT + i) + lane;
        const int grp8 = u / (16 * D / 16), within = u % (16 * D / 16);
        const int row = 8 * grp8 + ((within & 31) >> 2);
        const int ch = 4 * (within >> 5) + ((within & 3) ^ ((row >> 2) & 3));
        st_goff[i] = (row * D + ch * 8) * 2;
    }
    // (the host pass of hipcc silently drops a kernel whose body names this builtin; the calls
    // only exist in the device pass)
    // piece i of a tile's fetch: even i this wave's KiB piece i/2 of Q, odd i that of dO
    // (i < 2 LPT); piece 2 LPT the row statistics (wave 0)
    auto fetch_piece = [&](int tile, int buf, auto i_) __attribute__((always_inline)) {
#if defined(__HIP_DEVICE_COMPILE__)
        constexpr int i = decltype(i_)::value;
        if constexpr (i < 2 * LPT && i % 2 == 0) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(qrs, (lds_void*)(lds + buf * TILE_B + 1024 * (wave * LPT + i / 2)),
                                                     16, st_goff[i / 2], tile * TILE_B, 0, 0);
        } else if constexpr (i < 2 * LPT) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(ors, (lds_void*)(lds + (3 + buf) * TILE_B + 1024 * (wave * LPT + i / 2)),
                                                     16, st_goff[i / 2], tile * TILE_B, 0, 0);
        } else if (wave == 0) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(lrs, (lds_void*)(stat + buf * 2 * kTile), 4, lane * 4, tile * kTile * 4, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(drs, (lds_void*)(stat + buf * 2 * kTile + kTile), 4, lane * 4, tile * kTile * 4, 0, 0);
        }
#endif
    };
    auto fetch_tile = [&](int tile, int buf) __attribute__((always_inline)) {
        static_for<2 * LPT + 1>([&](auto ic) __attribute__((always_inline)) { fetch_piece(tile, buf, ic); });
    };

    f32x16 dk[DB], dv[DB];
#pragma unroll
    for (int db = 0; db < DB; ++db)
#pragma unroll
        for (int i = 0; i < 16; ++i) { dk[db][i] = 0.f; dv[db][i] = 0.f; }

    // The tiles of queries this key block sees, in the order they are taken: tiles that need the
    // mask first (the ragged last tile, then under the causal mask the two tiles that straddle the
    // diagonal; every tile if the key block itself is ragged), then the others.  The two kinds
    // run in two loops so that the common one carries no mask code.
    const int T_all = (N + kTile - 1) / kTile;
    const int t0 = CAUSAL ? k0 / kTile : 0;        // causal: queries before the block's first key see none of it
    const int n_tiles = T_all - t0;
    const int has_tail = (N % kTile) != 0 ? 1 : 0;
    const int n_edge = k0 + BK > N ? n_tiles : has_tail + (CAUSAL ? min(2, n_tiles - has_tail) : 0);
    auto tile_of = [=](int j) __attribute__((always_inline)) { return (has_tail && j == 0) ? T_all - 1 : t0 + j - has_tail; };
    fetch_tile(tile_of(0), 0);
    fetch_tile(tile_of(1), 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int cur = 0, nxt = 1, fre = 2;                 // buffers of tile j, j + 1, j + 2

    // Per-lane bases of the fragment reads; everything else in an address is an immediate.
    //   row read of block qb, k-step ks: row 32qb + r, chunk 2ks + h
    //   transposing read of (qb, st, db): rows 32qb + 16st + 8hi + 4h + tq, chunk 4db + 2tg1 + (tp >> 1)
    constexpr int SUB_B = 16 * D;
    const int tq = (lane & 15) >> 2, tp = lane & 3, tg1 = (lane >> 4) & 1;
    int rbase[2], tbase[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        rbase[e] = SUB_B * (r >> 3) + 64 * (r & 7) + 16 * ((2 * e + h) ^ ((r >> 2) & 3));
        tbase[e] = 64 * (4 * h + tq) + 16 * ((2 * tg1 + (tp >> 1)) ^ (2 * e + h)) + 8 * (tp & 1) + SUB_B * e;
    }
    using Plan = DkvPlanFor<KS>;
    static_assert(Plan::plan.ok && 4 * DB == 2 * KS, "every vector operation placed before its deadline");
    // FW: fragments requested ahead of the MFMA stream.  The tile's last FW steps request none, and
    // that is where the next-but-one tile's loads are issued, one piece every PSTEP steps.
    constexpr int NSTEP = 8 * KS, FW = 8, PSTEP = FW / (2 * LPT);
    static_assert(PSTEP >= 1 && PSTEP * 2 * LPT == FW, "the tile's loads fill its last FW steps");

    auto run_tiles = [&](int j0, int j1, auto edge_) __attribute__((always_inline)) {
    for (int j = j0; j < j1; ++j) {
        const int qt0 = tile_of(j) * kTile;
        const char* Qb = lds + cur * TILE_B;
        const char* dOb = lds + (3 + cur) * TILE_B;
        const float* lse2b = stat + cur * 2 * kTile;
        const float* ndelb = lse2b + kTile;

        // S = Q K^T and dP = dO V^T put the key on the lane; accumulator register i of block qb is
        // query 32qb + (i & 3) + 8 (i >> 2) + 4h, so the rows' lse2 come as four broadcast float4
        // reads per block, and four more start the dP accumulators at -delta.  S and dP live in
        // VECTOR registers (acc_v / first_v), dK and dV in accumulator registers.
        f32x16 s[2], dp[2];
        f32x4 l4[2][4];
        vec8 pb[4], sb[4];
#pragma unroll
        for (int qb = 0; qb < 2; ++qb)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                l4[qb][g] = *reinterpret_cast<const f32x4*>(lse2b + 32 * qb + 8 * g + 4 * h);
                const f32x4 nd = *reinterpret_cast<const f32x4*>(ndelb + 32 * qb + 8 * g + 4 * h);
                dp[qb][4 * g] = nd[0]; dp[qb][4 * g + 1] = nd[1]; dp[qb][4 * g + 2] = nd[2]; dp[qb][4 * g + 3] = nd[3];
            }

        // pairs (query, key) outside the problem or above the diagonal: element c = 32qb + 8(i>>2) + (i&3)
        // of this lane is masked iff c >= c_hi or c < c_lo
        const int c_hi = krow < N ? N - (qt0 + 4 * h) : -1;
        const int c_lo = CAUSAL ? krow - (qt0 + 4 * h) : -(1 << 30);

        // LDS fragment of step g: A, B: a row read of Q (S) or dO (dP); C, D: a transposing read of
        // dO (dV) or Q (dK), k-step st = i / (2 DB), then (db, which)
        auto frag = [&](auto gc) __attribute__((always_inline)) -> s16x8 {
            constexpr int g = decltype(gc)::value;
            if constexpr (g < 4 * KS) {
                constexpr int qb = g / (2 * KS), which = (g / KS) & 1, ks = g % KS;
                const char* tile = which ? dOb : Qb;
                return *reinterpret_cast<const s16x8*>(tile + rbase[ks & 1] + SUB_B * 4 * qb + 512 * (ks >> 1));
            } else {
                constexpr int qb = (g - 4 * KS) / (2 * KS), i = g % (2 * KS);
                constexpr int st = i / (2 * DB), db = (i / 2) % DB, which = i & 1;
                const char* tile = which ? Qb : dOb;
                constexpr int imm = SUB_B * 2 * (2 * qb + st) + 512 * db;
                const s16x4 lo = lds_read_tr(tile + tbase[0] + imm);
                const s16x4 hi = lds_read_tr(tile + tbase[1] + imm);
                s16x8 a;
                a[0] = lo[0]; a[1] = lo[1]; a[2] = lo[2]; a[3] = lo[3];
                a[4] = hi[0]; a[5] = hi[1]; a[6] = hi[2]; a[7] = hi[3];
                return a;
            }
        };
        auto mma = [&](auto gc, s16x8 a) __attribute__((always_inline)) {
            constexpr int g = decltype(gc)::value;
            if constexpr (g < 4 * KS) {
                constexpr int qb = g / (2 * KS), which = (g / KS) & 1, ks = g % KS;
                if constexpr (which) M::acc_v(a, vf[ks], dp[qb]);
                else if constexpr (ks == 0) M::first_v(a, kf[0], s[qb]);
                else M::acc_v(a, kf[ks], s[qb]);
            } else {
                constexpr int qb = (g - 4 * KS) / (2 * KS), i = g % (2 * KS);
                constexpr int st = i / (2 * DB), db = (i / 2) % DB, which = i & 1;
                if constexpr (which) dk[db] = M::run(a, sb[2 * qb + st], dk[db]);
                else dv[db] = M::run(a, pb[2 * qb + st], dv[db]);
            }
        };
        // one planned vector operation
        auto vop = [&](auto edge_, auto op_) __attribute__((always_inline)) {
            constexpr bool EDGE = decltype(edge_)::value != 0;
            constexpr int o = decltype(op_)::value;
            constexpr int qb = Plan::plan.op[o].blk, kind = Plan::plan.op[o].kind, e = Plan::plan.op[o].e;
            if constexpr (kind == 0) {
                s[qb][e] = __builtin_fmaf(s[qb][e], sl2, -l4[qb][e >> 2][e & 3]);
            } else if constexpr (kind == 1) {
                const float pe = __builtin_amdgcn_exp2f(s[qb][e]);
                constexpr int c = 32 * qb + 8 * (e >> 2) + (e & 3);
                if constexpr (EDGE) s[qb][e] = (c >= c_hi || c < c_lo) ? 0.f : pe;
                else s[qb][e] = pe;
            } else if constexpr (kind == 2) {
                dp[qb][e] *= s[qb][e];
            } else if constexpr (kind == 3) {
                pb[2 * qb + (e >> 2)][(2 * e) & 7] = (T)s[qb][2 * e];
                pb[2 * qb + (e >> 2)][((2 * e) & 7) + 1] = (T)s[qb][2 * e + 1];
            } else {
                sb[2 * qb + (e >> 2)][(2 * e) & 7] = (T)dp[qb][2 * e];
                sb[2 * qb + (e >> 2)][((2 * e) & 7) + 1] = (T)dp[qb][2 * e + 1];
            }
        };
        // The tile as ONE software-pipelined stream of 8 KS MFMAs: the LDS fragment of MFMA g + FW is
        // requested when MFMA g issues (nothing else hides the LDS latency), the planned vector
        // operations of step g follow in that MFMA's shadow, and the order is pinned.
        s16x8 win[FW];
        static_for<FW>([&](auto ic) __attribute__((always_inline)) { win[decltype(ic)::value] = frag(ic); });
        static_for<NSTEP>([&](auto gc) __attribute__((always_inline)) {
            constexpr int g = decltype(gc)::value;
            const s16x8 a = win[g % FW];
            mma(gc, a);
            if constexpr (g + FW < NSTEP) win[g % FW] = frag(IC<g + FW>{});
            // Tile j + 2 into the buffer tile j - 1 was read from (everybody left it at the last
            // barrier; unconditional, past the end a tile nobody reads), spread over the tile's last
            // FW steps.  Two facts set this placement.  All four waves run in lock step, and the
            // 34 load instructions of a tile issued back to back at the top of the tile took the
            // one texture path of the CU some 900 cycles, a quarter of the tile, with every wave
            // held at its loads and every matrix core idle.  And hipcc makes each transposing LDS
            // read wait for ALL LDS-DMA loads in flight, so loads may only be issued where no such
            // read follows soon: behind the tile's last fragment request.  The wait in front
            // confirms this wave's pieces of tile j + 1, issued a tile ago.
            if constexpr (g == NSTEP - FW) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if constexpr (g >= NSTEP - FW && (g - (NSTEP - FW)) % PSTEP == 0)
                fetch_piece(tile_of(j + 2), fre, IC<(g - (NSTEP - FW)) / PSTEP>{});
            if constexpr (g == NSTEP - 1) fetch_piece(tile_of(j + 2), fre, IC<2 * LPT>{});
            constexpr int o0 = Plan::plan.start[g], o1 = Plan::plan.start[g + 1];
            static_for<o1 - o0>([&](auto oc) __attribute__((always_inline)) { vop(edge_, IC<o0 + decltype(oc)::value>{}); });
            __builtin_amdgcn_sched_barrier(0);
        });

        // tile j + 1 complete (all waves' pieces), tile j released.  The bare barrier: __syncthreads()
        // would first wait for the loads of tile j + 2 issued a moment ago.
        asm volatile("" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        const int was = cur;
        cur = nxt; nxt = fre; fre = was;
    }
    };
    run_tiles(0, n_edge, IC<1>{});
    run_tiles(n_edge, n_tiles, IC<0>{});

    if (krow < N) {
        T* krow_p = dKg + hoff + (size_t)krow * D;
        T* vrow_p = dVg + hoff + (size_t)krow * D;
#pragma unroll
        for (int db = 0; db < DB; ++db)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                vec4 wk, wv;
#pragma unroll
                for (int j = 0; j < 4; ++j) { wk[j] = (T)(dk[db][4 * g + j] * scale); wv[j] = (T)dv[db][4 * g + j]; }
                *reinterpret_cast<vec4*>(krow_p + 32 * db + 8 * g + 4 * h) = wk;
                *reinterpret_cast<vec4*>(vrow_p + 32 * db + 8 * g + 4 * h) = wv;
            }
    }
}

template <typename T, int D>
int launch_bwd(const void* q, const void* k, const void* v, const void* o, const float* lse, const void* dout,
               void* dq, void* dk, void* dv, float* delta, int64_t heads, int64_t N, float scale,
               bool causal, hipStream_t s) {
    const T* Q = static_cast<const T*>(q);
    const T* K = static_cast<const T*>(k);
    const T* V = static_cast<const T*>(v);
    const T* O = static_cast<const T*>(o);
    const T* dO = static_cast<const T*>(dout);
    const int64_t rows = heads * N;
    constexpr int RPB = 256 / (D / 8);
    // every size check comes before the first launch: a launch status must not be left behind
    // by an early return, to be reported against some later call of this thread
    const int64_t nqblk = cdiv(N, 256), nkblk = cdiv(N, 128);
    const int64_t nq = heads * nqblk, nk = heads * nkblk;
    if (nq > 0x7fffffffLL || nk > 0x7fffffffLL || cdiv(rows, RPB) > 0x7fffffffLL)
        return fail(MYFA_E_UNSUPPORTED, "attention backward: too many workgroups");
    launch_kernel((attn_bwd_delta_kernel<T, D>), dim3((unsigned)cdiv(rows, RPB)), dim3(256), 0, s, O, dO, lse, delta, (long long)rows);
    const float* lse2 = delta + rows;

    if (causal) {
        launch_kernel((attn_bwd_dkv_kernel<T, D, true>), dim3((unsigned)nk), dim3(256), 0, s, Q, K, V, dO, lse2, delta,
                           static_cast<T*>(dk), static_cast<T*>(dv), (int)N, (int)nkblk, (int)nk, scale);
        launch_kernel((attn_bwd_dq_kernel<T, D, true>), dim3((unsigned)nq), dim3(512), 0, s, Q, K, V, dO, lse, delta,
                           static_cast<T*>(dq), (int)N, (int)nqblk, (int)nq, scale);
    } else {
        launch_kernel((attn_bwd_dkv_kernel<T, D, false>), dim3((unsigned)nk), dim3(256), 0, s, Q, K, V, dO, lse2, delta,
                           static_cast<T*>(dk), static_cast<T*>(dv), (int)N, (int)nkblk, (int)nk, scale);
        launch_kernel((attn_bwd_dq_kernel<T, D, false>), dim3((unsigned)nq), dim3(512), 0, s, Q, K, V, dO, lse, delta,
                           static_cast<T*>(dq), (int)N, (int)nqblk, (int)nq, scale);
    }
    return check_launch("attn_bwd_mfma");
}

// ---- the allocation-free entry point's workspace -------------------------------------------
// A file-scope device array takes the place of the caller's `delta_ws`.  Every GPU the library
// runs on holds its own copy (8 MiB) from the moment the code object is loaded there until the
// process ends.  The array's name is only a handle on the host: its address on a device comes
// from hipGetSymbolAddress, asked once per device with that device current, and that pointer is
// what the kernels receive.
constexpr int64_t kScratchFloats = 2 * 1048576;   // 2 * B*H*N floats, B*H*N <= 2^20
static __device__ __align__(256) float scratch[kScratchFloats];

constexpr int kMaxDevices = 64;
std::mutex g_scratch_mu;
float* g_scratch_ptr[kMaxDevices] = {};

// Device address of `scratch` on the current device (nullptr + error message on failure).
float* scratch_on_current_device() {
    int dev = -1;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess || dev < 0 || dev >= kMaxDevices) {
        fail(MYFA_E_DEVICE, "attention backward (static workspace): no usable current device (%s)", hipGetErrorString(e));
        return nullptr;
    }
    std::lock_guard<std::mutex> lock(g_scratch_mu);
    if (g_scratch_ptr[dev] == nullptr) {
        void* p = nullptr;
        e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
        if (e != hipSuccess || p == nullptr) {
            fail(MYFA_E_DEVICE, "attention backward (static workspace): hipGetSymbolAddress on device %d: %s", dev, hipGetErrorString(e));
            return nullptr;
        }
        g_scratch_ptr[dev] = static_cast<float*>(p);
    }
    return g_scratch_ptr[dev];
}

}  // namespace

// The backward with the [2, heads, N] float workspace in the library's own device array instead
// of a caller buffer.  Nothing in the array has to start at zero: the delta kernel writes all
// 2 * heads * N words before the other two kernels read them.  Calls that overlap on one GPU
// (different streams) share the array and race.
int attn_bwd_mfma(const void* q, const void* k, const void* v, const void* o, const float* lse, const void* dout,
                  void* dq, void* dk, void* dv, float* delta, int64_t heads, int64_t N, int64_t d,
                  float scale, int dtype, bool causal, hipStream_t s);
int attn_bwd_mfma_static_ws(const void* q, const void* k, const void* v, const void* o, const float* lse, const void* dout,
                            void* dq, void* dk, void* dv, int64_t heads, int64_t N, int64_t d,
                            float scale, int dtype, bool causal, hipStream_t s) {
    if (heads > kScratchFloats / 2 || N > kScratchFloats / 2 || 2 * heads * N > kScratchFloats)
        return fail(MYFA_E_UNSUPPORTED,
                    "attention backward (static workspace): B*H*N = %lld exceeds the built-in workspace of %lld rows; "
                    "use myfa_attn_bwd_bhnd with a caller workspace",
                    (long long)(heads * N), (long long)(kScratchFloats / 2));
    float* ws = scratch_on_current_device();
    if (ws == nullptr) return MYFA_E_DEVICE;
    return attn_bwd_mfma(q, k, v, o, lse, dout, dq, dk, dv, ws, heads, N, d, scale, dtype, causal, s);
}

int attn_bwd_wide(const void* q, const void* k, const void* v, const void* o, const float* lse, const void* dout,
                  void* dq, void* dk, void* dv, float* delta, int64_t heads, int64_t N, int64_t d,
                  float scale, int dtype, bool causal, hipStream_t s);   // attn_bwd_wide.hip
int attn_bwd_f32(const float* q, const float* k, const float* v, const float* o, const float* lse, const float* dout,
                 float* dq, float* dk, float* dv, float* delta, int64_t heads, int64_t N, int64_t d,
                 float scale, bool causal, hipStream_t s);               // attn_bwd_f32.hip

// dtype: MYFA_F16 or MYFA_BF16; d: 64, 128 (this file's kernels), 192 or 256 (attn_bwd_wide.hip);
// or MYFA_F32 with any d up to 512 (attn_bwd_f32.hip).  heads = B*H.  `delta` is a caller-provided float32 workspace of 2 * heads * N elements (the
// library allocates nothing).
int attn_bwd_mfma(const void* q, const void* k, const void* v, const void* o, const float* lse, const void* dout,
                  void* dq, void* dk, void* dv, float* delta, int64_t heads, int64_t N, int64_t d,
                  float scale, int dtype, bool causal, hipStream_t s) {
    if (!(scale > 0.f)) return fail(MYFA_E_UNSUPPORTED, "attention backward: scale must be > 0, got %g", (double)scale);
    if (N > 0x7fffffffLL / 256) return fail(MYFA_E_UNSUPPORTED, "attention backward: N=%lld too large", (long long)N);
    if (dtype == MYFA_F32)
        return attn_bwd_f32(static_cast<const float*>(q), static_cast<const float*>(k), static_cast<const float*>(v),
                            static_cast<const float*>(o), lse, static_cast<const float*>(dout), static_cast<float*>(dq),
                            static_cast<float*>(dk), static_cast<float*>(dv), delta, heads, N, d, scale, causal, s);
    if ((dtype == MYFA_BF16 || dtype == MYFA_F16) && (d == 192 || d == 256))
        return attn_bwd_wide(q, k, v, o, lse, dout, dq, dk, dv, delta, heads, N, d, scale, dtype, causal, s);
    if (dtype == MYFA_BF16 && d == 128) return launch_bwd<__bf16, 128>(q, k, v, o, lse, dout, dq, dk, dv, delta, heads, N, scale, causal, s);
    if (dtype == MYFA_BF16 && d == 64)  return launch_bwd<__bf16, 64>(q, k, v, o, lse, dout, dq, dk, dv, delta, heads, N, scale, causal, s);
    if (dtype == MYFA_F16 && d == 128)  return launch_bwd<_Float16, 128>(q, k, v, o, lse, dout, dq, dk, dv, delta, heads, N, scale, causal, s);
    if (dtype == MYFA_F16 && d == 64)   return launch_bwd<_Float16, 64>(q, k, v, o, lse, dout, dq, dk, dv, delta, heads, N, scale, causal, s);
    return fail(MYFA_E_UNSUPPORTED, "attention backward: needs float16 or bfloat16 with head dim 64, 128, 192 or 256, or float32 with head dim up to 512 (got dtype %d, d %lld)", dtype, (long long)d);
}

}  // namespace myfa
