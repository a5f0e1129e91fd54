// Fused attention forward for f16 / bf16 with head dims above 128, up to 256.
//
//     O[b,h] = softmax(scale * Q[b,h] K[b,h]^T) V[b,h]      Q,K,V,O: [B,H,N,d], d a multiple of 8 in 136..256
//
// Same math as the narrower builds (the online-softmax recurrence of tests/test_flash_attn.py:82-93
// on v_mfma_f32_32x32x16, S^T = K Q^T so that a query's logits sit in one lane pair, O^T = V^T P^T
// with the transposing LDS read); what differs is the budget.  At 256 columns a wave's O^T tile
// alone is 128 accumulator registers and its Q fragments 64 more, so a wave needs the whole
// register file of its SIMD:
//   workgroup = 4 waves = 128 query rows, one wave per SIMD (512 registers), wave w owns rows 32w..32w+31;
//   K / V tiles of 32 keys (16 KiB at 256 columns) go HBM/L2 -> LDS directly (buffer_load ... lds:
//   rows past N and columns past d read as zeros, so any multiple of 8 up to the build's width
//   runs on it), three LDS buffers per tensor, the fetch two tiles ahead, one barrier per tile;
//   with nobody else on the SIMD the wave overlaps its own work: an iteration issues
//   S(t+1) = K(t+1) Q^T, looks for the tile's maxima, and then runs the weights of tile t+1 (exp2,
//   row sum) one element at a time in the shadow of the MFMAs of O^T += V(t)^T P(t)^T.
// The running max moves only in a tile where some row's max grew by more than `thr` (log2 units;
// P may reach 2^thr), which keeps the rescale of O^T out of most tiles; when it moves, O^T is
// rescaled behind the tile's last MFMA (P(t) was formed against the old max, like all of O^T).
// Builds: 192 and 256 columns.
#include "attn_mfma_common.h"

namespace myfa {
namespace {

constexpr int kThreadsW = 256;       // 4 waves
constexpr int kBKVW = 32;            // keys per tile
constexpr int kQBW = 2;              // 32-row query blocks per wave
// A buffer offset no tensor reaches (the launcher keeps a head below it) and that a tile offset
// added to it cannot wrap: fetches from there read zero without touching memory.
constexpr int kPastEndW = 0x7FFF0000;

// QB: 32-row query blocks per wave (a K or V fragment read from LDS feeds QB MFMAs)
template <typename T, int D, bool CAUSAL, int QB>
__global__ void __launch_bounds__(kThreadsW, 1)
attn_fwd_mfma_wide_kernel(const T* __restrict__ Qg, const T* __restrict__ Kg, const T* __restrict__ Vg,
                          T* __restrict__ Og, float* __restrict__ LSEg, int N, int nqblk, int nwg,
                          float sl2, float thr, int gd) {
    using M = Mma<T>;
    using vec8 = typename M::vec8;
    using vec4 = typename M::vec4;
    constexpr int TR = kBKVW;
    constexpr int RW = 32 * QB, kBQW = 4 * RW;    // query rows per wave, per workgroup
    constexpr int TILE_B = TR * D * 2;            // bytes per K (or V) tile image
    constexpr int KS = D / 16;                    // k-steps of S^T = K Q^T
    constexpr int DB = D / 32;                    // 32-row blocks of O^T
    constexpr int LPT = TR * (D / 8) / kThreadsW; // 1-KiB pieces per wave and tile
    static_assert(LPT * kThreadsW == TR * (D / 8), "whole 1-KiB pieces per wave");

    __shared__ __attribute__((aligned(16))) char lds[6 * TILE_B];   // K0 K1 K2 V0 V1 V2

    // Workgroups that share one head's K/V get consecutive logical ids, and each XCD
    // (blockIdx % 8 under round-robin dispatch) a contiguous range of them.  Speed only.
    int lid;
    {
        const int bid = blockIdx.x, xcd = bid & 7, idx = bid >> 3, qq = nwg >> 3, rr = nwg & 7;
        lid = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + idx;
    }
    const int head = lid / nqblk;
    // causal: the long query blocks (high rows) of each head are started first
    const int qblk = CAUSAL ? nqblk - 1 - (lid - head * nqblk) : lid - head * nqblk;
    const int q0 = qblk * kBQW;

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;

    // gd <= D is the head dim in memory: rows are gd elements apart, chunks past it read as zeros
    const size_t hoff = (size_t)head * (size_t)N * gd;
    const unsigned head_bytes = (unsigned)N * (unsigned)gd * 2u;
    [[maybe_unused]] const __amdgpu_buffer_rsrc_t qrs = __builtin_amdgcn_make_buffer_rsrc((void*)(Qg + hoff), 0, head_bytes, 0x00020000);
    [[maybe_unused]] const __amdgpu_buffer_rsrc_t krs = __builtin_amdgcn_make_buffer_rsrc((void*)(Kg + hoff), 0, head_bytes, 0x00020000);
    [[maybe_unused]] const __amdgpu_buffer_rsrc_t vrs = __builtin_amdgcn_make_buffer_rsrc((void*)(Vg + hoff), 0, head_bytes, 0x00020000);

    // ---- Q^T fragments: lane (r, h) holds Q[row][16ks + 8h .. +7] for ks = 0..KS-1 (rows past N: zeros)
    const int qrow = q0 + wave * RW + r;          // block qb: + 32 qb
    vec8 qf[QB][KS];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int ch = 2 * ks + h;
            [[maybe_unused]] const int off = ch * 8 < gd ? (qrow + 32 * qb) * gd * 2 + ch * 16 : kPastEndW;
#if defined(__HIP_DEVICE_COMPILE__)
            qf[qb][ks] = __builtin_bit_cast(vec8, __builtin_amdgcn_raw_buffer_load_b128(qrs, off, 0, 0));
#endif
        }

    // ---- K / V tiles: piece i of wave w is image bytes 1024 (w LPT + i) .. + 1023, lane L at + 16 L,
    // so the image (attn_mfma_common.h) is produced by choosing each lane's SOURCE chunk
    int goff[LPT];
#pragma unroll
    for (int i = 0; i < LPT; ++i) {
        const int u = 64 * (wave * LPT + i) + lane;
        const int grp8 = u / D, within = u % D;          // 16-byte units: D of them per 8-row group
        const int row = 8 * grp8 + ((within & 31) >> 2);
        const int ch = 4 * (within >> 5) + ((within & 3) ^ ((row >> 2) & 3));
        goff[i] = ch * 8 < gd ? (row * gd + ch * 8) * 2 : kPastEndW;
    }
    [[maybe_unused]] const int tile_stride = TR * gd * 2;          // bytes of a tile in memory
    // piece p of a fetch (p < 2 LPT): even p this wave's piece p/2 of K tile `kt` into K buffer `kbuf`,
    // odd p that of V tile `vt` into V buffer `vbuf`.  (The host pass of hipcc silently drops a
    // kernel whose body names this builtin.)
    auto fetch_piece = [&](int kt, int kbuf, int vt, int vbuf, int p) __attribute__((always_inline)) {
#if defined(__HIP_DEVICE_COMPILE__)
        if ((p & 1) == 0)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(krs, (lds_void*)(lds + kbuf * TILE_B + 1024 * (wave * LPT + p / 2)),
                                                     16, goff[p / 2], kt * tile_stride, 0, 0);
        else
            __builtin_amdgcn_raw_ptr_buffer_load_lds(vrs, (lds_void*)(lds + (3 + vbuf) * TILE_B + 1024 * (wave * LPT + p / 2)),
                                                     16, goff[p / 2], vt * tile_stride, 0, 0);
#endif
    };

    // ---- fragment addresses (per-lane bases; everything else is an immediate)
    const int tq = (lane & 15) >> 2, tp = lane & 3, tg1 = (lane >> 4) & 1;
    int rbase[2], tbase[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        rbase[e] = 16 * D * (r >> 3) + 64 * (r & 7) + 16 * ((2 * e + h) ^ ((r >> 2) & 3));
        tbase[e] = 64 * (4 * h + tq) + 16 * ((2 * tg1 + (tp >> 1)) ^ (2 * e + h)) + 8 * (tp & 1) + 16 * D * e;
    }
    // The fragment reads are written out as instructions, with their waits (lgkmcnt: LDS answers one
    // wave's requests in order, so "all but the n youngest" is exact).  Left to the compiler, every LDS
    // read that follows an LDS-DMA fetch first waits for ALL fetches in flight (it cannot tell the
    // buffers apart), and a fetch then has one MFMA stream to arrive in instead of a whole iteration.
    // What orders reads and fetches here: a buffer is fetched into only behind the barrier that
    // follows its last read, and read only behind the barrier that follows the fetching waves' wait.
    const unsigned lds0 = (unsigned)(uintptr_t)(lds_void*)lds;
    auto k_frag = [&]([[maybe_unused]] unsigned a0, [[maybe_unused]] unsigned a1, auto ks_) __attribute__((always_inline)) -> s16x8 {
        [[maybe_unused]] constexpr int ks = decltype(ks_)::value;
        s16x8 a = {};          // (the host pass sees no asm; the device pass overwrites it)
#if defined(__HIP_DEVICE_COMPILE__)
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(a) : "v"((ks & 1) ? a1 : a0), "n"(512 * (ks >> 1)));
#endif
        return a;
    };
    auto v_frag = [&]([[maybe_unused]] unsigned a0, [[maybe_unused]] unsigned a1, auto i_) __attribute__((always_inline)) -> s16x8 {
        [[maybe_unused]] constexpr int i = decltype(i_)::value, st = i / DB, db = i % DB;       // 16 keys x 32 columns
        s16x4 lo, hi;
#if defined(__HIP_DEVICE_COMPILE__)
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(lo) : "v"(a0), "n"(16 * D * 2 * st + 512 * db));
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(hi) : "v"(a1), "n"(16 * D * 2 * st + 512 * db));
#endif
        s16x8 a;
        a[0] = lo[0]; a[1] = lo[1]; a[2] = lo[2]; a[3] = lo[3];
        a[4] = hi[0]; a[5] = hi[1]; a[6] = hi[2]; a[7] = hi[3];
        return a;
    };
    auto lgkm_wait = [&](auto n_) __attribute__((always_inline)) {
        asm volatile("s_waitcnt lgkmcnt(%0)" :: "n"(decltype(n_)::value));
    };

    f32x16 o[QB][DB];
    float m[QB], l[QB];    // running (deferred) row max of scale*log2(e)*S; this lane half's share of the row sum
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
#pragma unroll
        for (int db = 0; db < DB; ++db)
#pragma unroll
            for (int i = 0; i < 16; ++i) o[qb][db][i] = 0.f;
        m[qb] = -INFINITY;
        l[qb] = 0.f;
    }

    const int T_all = (N + TR - 1) / TR;
    const int ntiles = CAUSAL ? min(T_all, (q0 + kBQW) / TR) : T_all;
    // the last key a row may see: tile 0 always holds key 0, which every row sees
    auto key_limit = [&](int qb) __attribute__((always_inline)) { return CAUSAL ? min(N - 1, qrow + 32 * qb) : N - 1; };
    auto tile_or_end = [&](int t) __attribute__((always_inline)) { return t < ntiles ? t : T_all; };   // past the end: zeros

    // prologue: K(0), V(0), K(1) -- and V(1), K(2) in flight behind them
#pragma unroll
    for (int p = 0; p < 2 * LPT; ++p) fetch_piece(0, 0, 0, 0, p);
#pragma unroll
    for (int p = 0; p < 2 * LPT; ++p) fetch_piece(tile_or_end(1), 1, tile_or_end(1), 1, p);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
#pragma unroll
    for (int p = 0; p < 2 * LPT; p += 2) fetch_piece(tile_or_end(2), 2, 0, 0, p);      // K(2) only

    constexpr int FW = 4;                          // LDS fragments requested ahead of the MFMA stream
    constexpr int NPV = 2 * DB;                    // steps (fragments) of O^T += V^T P^T per tile
    constexpr int EP = (16 * QB + NPV - 1) / NPV;  // weights formed per step of that stream

    f32x16 s[QB];
    // S^T and O^T come from MFMAs the compiler does not see as such (Mma::acc_v, acc_a): whoever reads
    // them with a vector instruction first waits out the last one's result latency (8 passes of 4 cycles
    // and change)
    auto mfma_drain = [&]() __attribute__((always_inline)) {
        asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");
    };
    // S^T of one tile: key (i & 3) + 8 (i >> 2) + 4 h of the tile in register i
    // `after(i)` runs between the two MFMAs of step i (the loop hangs the tile fetches there: the
    // stream has issue slots to spare, the other one's are taken by the weights)
    // One window of FW fragments serves both streams: as the S stream's last FW steps free its slots,
    // the first fragments of the O^T stream are requested into them (NEXT), so that they arrive under
    // the search for the maxima instead of in front of that stream's first MFMA (+0.6 % on random
    // operands, +1.8 % in cycles).  (Moving the previous tile's conversions between the S stream's
    // first requests and its first MFMA, to cover that latency too, was measured: -3 %, -7 % in cycles.)
    static_assert(KS % FW == 0 && NPV >= FW, "fragment j of either stream lives in window slot j % FW");
    s16x8 win[FW];
    auto s_prefill = [&](int kbuf) __attribute__((always_inline)) {
        const unsigned a0 = lds0 + kbuf * TILE_B + rbase[0], a1 = lds0 + kbuf * TILE_B + rbase[1];
        static_for<FW>([&](auto ic) __attribute__((always_inline)) { win[decltype(ic)::value] = k_frag(a0, a1, ic); });
    };
    auto s_run = [&](int kbuf, auto next_, int vbuf, auto&& after) __attribute__((always_inline)) {
        constexpr bool NEXT = decltype(next_)::value != 0;
        const unsigned a0 = lds0 + kbuf * TILE_B + rbase[0], a1 = lds0 + kbuf * TILE_B + rbase[1];
        [[maybe_unused]] const unsigned va0 = lds0 + (3 + vbuf) * TILE_B + tbase[0], va1 = lds0 + (3 + vbuf) * TILE_B + tbase[1];
        static_for<KS>([&](auto ic) __attribute__((always_inline)) {
            constexpr int i = decltype(ic)::value;
            // requests younger than fragment i: the K fragments behind it, and (two reads each) the
            // V fragments the tail has asked for so far
            constexpr int j = i - (KS - FW);
            lgkm_wait(IC<(NEXT && j >= 0 ? FW - 1 + j : (KS - 1 - i < FW - 1 ? KS - 1 - i : FW - 1))>{});
            const s16x8 a = win[i % FW];
#pragma unroll
            for (int qb = 0; qb < QB; ++qb) {
                // pinned to vector registers (the accumulator file belongs to O^T)
                if constexpr (i == 0) M::first_v(a, qf[qb][0], s[qb]);
                else M::acc_v(a, qf[qb][i], s[qb]);
                if (qb == 0) {
                    if constexpr (i + FW < KS) win[i % FW] = k_frag(a0, a1, IC<i + FW>{});
                    after(ic);
                }
            }
            // (behind the step's LAST MFMA: the slot is its operand)
            if constexpr (NEXT && j >= 0) win[i % FW] = v_frag(va0, va1, IC<(j >= 0 ? j : 0)>{});
            __builtin_amdgcn_sched_barrier(0);
        });
        mfma_drain();
    };
    // mask, tile maxima, and whether the running max moves (wave-uniform); returns the max the
    // tile's weights are formed against
    float m_new[QB];
    bool moved = false;
    auto look = [&](int kv0) __attribute__((always_inline)) {
        if (kv0 + TR > N || (CAUSAL && kv0 + TR - 1 > q0 + wave * RW)) {
#pragma unroll
            for (int qb = 0; qb < QB; ++qb)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int key = kv0 + (i & 3) + 8 * (i >> 2) + 4 * h;
                    if (key > key_limit(qb)) s[qb][i] = -INFINITY;
                }
        }
        float mt[QB];
        bool grew = false;
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) {
            float c0 = vmax3(s[qb][0], s[qb][1], s[qb][2]), c1 = vmax3(s[qb][3], s[qb][4], s[qb][5]);
            c0 = vmax3(c0, s[qb][6], s[qb][7]);
            c1 = vmax3(c1, s[qb][8], s[qb][9]);
            c0 = vmax3(c0, s[qb][10], s[qb][11]);
            c1 = vmax3(c1, s[qb][12], s[qb][13]);
            mt[qb] = half_exchange_max(vmax3(c0, c1, vmax2(s[qb][14], s[qb][15]))) * sl2;
            grew = grew || mt[qb] - m[qb] > thr;
        }
        moved = __builtin_amdgcn_ballot_w64(grew) != 0;   // always in tile 0 (m = -inf)
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) {
            m_new[qb] = m[qb];
            if (moved) m_new[qb] = fmaxf(m[qb], mt[qb]);
        }
    };
    float psum[QB];
    // weight e of the wave's 16 QB: block e % QB, register e / QB
    auto weight = [&](int e) __attribute__((always_inline)) {
        const int qb = e % QB, i = e / QB;
        s[qb][i] = __builtin_amdgcn_exp2f(__builtin_fmaf(s[qb][i], sl2, -m_new[qb]));
        psum[qb] = i == 0 ? s[qb][i] : psum[qb] + s[qb][i];
    };
    vec8 pb[QB][2];
    // behind the tile's last MFMA: O^T and l follow the max, the new weights become the B operand
    auto settle = [&](auto first_) __attribute__((always_inline)) {
        if constexpr (decltype(first_)::value != 0) {
            // tile 0: O^T and l are still zero, the max simply starts
#pragma unroll
            for (int qb = 0; qb < QB; ++qb) m[qb] = m_new[qb];
        } else if (moved) {
            mfma_drain();
#pragma unroll
            for (int qb = 0; qb < QB; ++qb) {
                const float alpha = __builtin_amdgcn_exp2f(m[qb] - m_new[qb]);
                l[qb] *= alpha;
#pragma unroll
                for (int db = 0; db < DB; ++db)
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        // through one vector register at a time: O^T stays in accumulator registers
                        // (left to the compiler the whole tile is copied out, scaled and copied back,
                        // and at 64 rows x 256 columns per wave that no longer fits)
                        float x = o[qb][db][i];
#if defined(__HIP_DEVICE_COMPILE__)
                        float tmp;
                        asm volatile("v_accvgpr_read_b32 %1, %0\n\tv_mul_f32 %1, %1, %2\n\tv_accvgpr_write_b32 %0, %1"
                                     : "+a"(x), "=&v"(tmp) : "v"(alpha));
#endif
                        o[qb][db][i] = x;
                    }
                m[qb] = m_new[qb];
            }
        }
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) {
            l[qb] += psum[qb];
#pragma unroll
            for (int st = 0; st < 2; ++st)
#pragma unroll
                for (int j = 0; j < 8; ++j) pb[qb][st][j] = (T)s[qb][8 * st + j];
        }
    };

    // tile 0 on its own: nothing to run under yet
    {
        s_prefill(0);
        s_run(0, IC<0>{}, 0, [](auto) __attribute__((always_inline)) {});
        look(0);
#pragma unroll
        for (int e = 0; e < 16 * QB; ++e) weight(e);
        settle(IC<1>{});
        // (K buffer 0 is fetched into again in the first iteration: everybody has to be through with it)
        __builtin_amdgcn_s_barrier();
    }

    // iteration t: K(t+1) in K buffer kc, V(t) in V buffer vc; the fetches of K(t+3) and V(t+2) go
    // into the buffers K(t) and V(t-1) were read from
    int kc = 1, kn = 2, kf = 0, vc = 0, vn = 1, vf = 2;
    for (int t = 0; t < ntiles; ++t) {
        const bool more = t + 1 < ntiles;          // uniform
        // the previous iteration's weights become this one's B operand (and O^T, l follow a moved max)
        // while the S stream's first fragments are on their way
        if (more) s_prefill(kc);
        if (more) {
            // The fetches of K(t+3), V(t+2) -- into the buffers K(t), V(t-1) were read from an iteration ago --
            // go out one piece every other step of this stream.  Nothing orders them against the
            // fragment reads but the barriers (the reads are asm, the compiler adds no wait).
            s_run(kc, IC<1>{}, vc, [&](auto ic) __attribute__((always_inline)) {
                constexpr int i = decltype(ic)::value;
                static_assert(2 * (2 * LPT) <= KS, "a piece every other step");
                if constexpr (i % 2 == 1 && i / 2 < 2 * LPT) fetch_piece(tile_or_end(t + 3), kf, tile_or_end(t + 2), vf, i / 2);
            });
            look((t + 1) * TR);
        }
        {
            const unsigned a0 = lds0 + (3 + vc) * TILE_B + tbase[0], a1 = lds0 + (3 + vc) * TILE_B + tbase[1];
            // (its first FW fragments were asked for in the S stream's tail, if there was one)
            if (!more) static_for<FW>([&](auto ic) __attribute__((always_inline)) { win[decltype(ic)::value] = v_frag(a0, a1, ic); });
            static_for<NPV>([&](auto ic) __attribute__((always_inline)) {
                constexpr int i = decltype(ic)::value;
                lgkm_wait(IC<2 * (NPV - 1 - i < FW - 1 ? NPV - 1 - i : FW - 1)>{});
                const s16x8 a = win[i % FW];
                // The wave issues in order and an MFMA holds the issue of the next one for its 32 cycles:
                // the step's other instructions are dealt BETWEEN its MFMAs, the fragment request and half
                // the weights behind the first, the rest behind the last (all of them behind the last
                // MFMA cost 16 cycles per step: they no longer fit one MFMA's shadow).
                // (Behind the last tile the weights work on stale registers and nothing reads the results.)
                constexpr int EA = (EP + QB - 1) / QB;    // weights per MFMA of the step
#pragma unroll
                for (int qb = 0; qb < QB; ++qb) {
                    M::acc_a(a, pb[qb][i / DB], o[qb][i % DB]);
                    if (qb == 0) {
                        if constexpr (i + FW < NPV) win[i % FW] = v_frag(a0, a1, IC<i + FW>{});
                    }
#pragma unroll
                    for (int e = EA * qb; e < EA * (qb + 1) && e < EP; ++e)
                        if (EP * i + e < 16 * QB) weight(EP * i + e);
                    if (qb + 1 < QB) __builtin_amdgcn_sched_barrier(0);
                }
                // Late in the stream: this wave's pieces of K(t+2) and V(t+1), issued an iteration ago, are
                // in (all but the 2 LPT fetches of this iteration; the loop holds no other vector-memory
                // operation -- tests/test_kernel_resources.py checks that it spills nothing).  Behind the
                // last tile's stream nothing was issued and nothing is waited for or needed.
                if constexpr (i == NPV - FW) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * LPT) : "memory");
                __builtin_amdgcn_sched_barrier(0);
            });
        }
        if (more) settle(IC<0>{});
        // K(t+2), V(t+1) complete (all waves' pieces), K(t+1), V(t) released; the bare barrier, not
        // __syncthreads(), which would first wait for the fetches issued a moment ago
        asm volatile("" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        const int kw = kc; kc = kn; kn = kf; kf = kw;
        const int vw = vc; vc = vn; vn = vf; vf = vw;
    }

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // no fetch may still be writing LDS when the wave ends
    // ---- epilogue: O = O^T / l, written as rows; lane (r,h) holds, for its query,
    // d = 32db + 8g + 4h + (0..3) in registers 4g..4g+3 of block db
    mfma_drain();
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        const float ltot = half_exchange_sum(l[qb]);
        const float inv = 1.0f / ltot;
        const int row = qrow + 32 * qb;
        if (row < N) {
            T* orow = Og + hoff + (size_t)row * gd;
#pragma unroll
            for (int db = 0; db < DB; ++db)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int col = 32 * db + 8 * g + 4 * h;
                    if (col < gd) {               // (gd is a multiple of 8: the four columns are in or out together)
                        vec4 w;
#pragma unroll
                        for (int j = 0; j < 4; ++j) w[j] = (T)(o[qb][db][4 * g + j] * inv);
                        *reinterpret_cast<vec4*>(orow + col) = w;
                    }
                }
            if (LSEg != nullptr && h == 0)
                LSEg[(size_t)head * N + row] = (m[qb] + __builtin_amdgcn_logf(ltot)) * 0.6931471805599453f;
        }
    }
}

template <typename T, int D, int QB>
int launch_wide(const void* q, const void* k, const void* v, void* o, float* lse,
                int64_t heads, int64_t N, int64_t d, float scale, float thr, bool causal, hipStream_t s) {
    // byte offsets inside a head are 32-bit, up to three tiles past its end
    if ((N + 4 * kBKVW) * d * 2 >= (int64_t)kPastEndW) return fail(MYFA_E_UNSUPPORTED, "attention: N*d=%lld per head too large", (long long)(N * d));
    const int64_t nqblk = cdiv(N, (int64_t)(128 * QB));
    const int64_t nwg = heads * nqblk;
    if (nwg > 0x7fffffffLL) return fail(MYFA_E_UNSUPPORTED, "attention: too many workgroups (%lld)", (long long)nwg);
    const float sl2 = scale * 1.4426950408889634f;
    auto go = [&](auto kern) {
        launch_kernel(kern, dim3((unsigned)nwg), dim3(kThreadsW), 0, s,
                      static_cast<const T*>(q), static_cast<const T*>(k), static_cast<const T*>(v),
                      static_cast<T*>(o), lse, (int)N, (int)nqblk, (int)nwg, sl2, thr, (int)d);
    };
    if (causal) go(attn_fwd_mfma_wide_kernel<T, D, true, QB>);
    else go(attn_fwd_mfma_wide_kernel<T, D, false, QB>);
    return check_launch("attn_fwd_mfma_wide");
}

}  // namespace

// dtype: MYFA_F16 or MYFA_BF16; d: a multiple of 8 in 136..256; heads = B*H; scale > 0
int attn_fwd_mfma_wide(const void* q, const void* k, const void* v, void* o, float* lse,
                       int64_t heads, int64_t N, int64_t d, float scale, float thr, int dtype, bool causal, hipStream_t s) {
    if (d <= 128 || d > 256 || d % 8 != 0)
        return fail(MYFA_E_UNSUPPORTED, "attention (f16/bf16, wide): head dim %lld is not a multiple of 8 in 136..256", (long long)d);
    if (dtype == MYFA_BF16)
        return d <= 192 ? launch_wide<__bf16, 192, kQBW>(q, k, v, o, lse, heads, N, d, scale, thr, causal, s)
                        : launch_wide<__bf16, 256, kQBW>(q, k, v, o, lse, heads, N, d, scale, thr, causal, s);
    if (dtype == MYFA_F16)
        return d <= 192 ? launch_wide<_Float16, 192, kQBW>(q, k, v, o, lse, heads, N, d, scale, thr, causal, s)
                        : launch_wide<_Float16, 256, kQBW>(q, k, v, o, lse, heads, N, d, scale, thr, causal, s);
    return fail(MYFA_E_INVALID, "attention (f16/bf16, wide): dtype tag %d", dtype);
}

}  // namespace myfa
