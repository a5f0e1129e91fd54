// Fused attention forward for f16 / bf16 on gfx950 matrix cores, v_mfma_f32_16x16x32 build.
//
//     O[b,h] = softmax(scale * Q[b,h] K[b,h]^T) V[b,h]      Q,K,V,O: [B,H,N,D], D in {64,128}
//
// The kernel the reference sketches and leaves as a TODO
// (csrc/my_flash_attn/flash_attention.cu:23-69); the math is the online-softmax recurrence of
// its Python restatement (tests/test_flash_attn.py:82-93).  Same work decomposition, LDS-DMA
// rings and two-phase ("ping-pong") schedule as attn_fwd_mfma_pp_kernel in attn_fwd_mfma.hip;
// what differs is the MFMA shape, and with it every register layout:
//
//   v_mfma_f32_16x16x32: D[16x16] += A[16x32] B[32x16].  Lane l = (i = l & 15, g = l >> 4)
//   holds A[row i][k = 8g..8g+7], B[k = 8g..8g+7][col i] and D[row 4g + r][col i], r = 0..3.
//
//   A wave owns 32 query rows as two blocks of 16 (qb = 0, 1); a K/V tile is 64 keys = four
//   blocks of 16 (kb).  S^T = K Q^T: A = 16 keys x 32 of d (a ds_read_b128 per lane), B = Q^T
//   from registers; every K fragment feeds two MFMAs (qb = 0, 1).  The accumulator
//   s[kb][qb][r] is S[query 16qb + i][key 16kb + 4g + r]: a lane holds 16 logits of each of TWO
//   queries, and a query's 64 logits are spread over the four lanes i, i+16, i+32, i+48.  Row
//   max / sum are in-lane loops plus one three-instruction exchange for both queries at once
//   (v_permlane32_swap, v_permlane16_swap, v_permlane32_swap).
//   O^T = V^T P^T: the accumulators of key blocks 2st and 2st+1, converted in place, are the B
//   operand of k-step st (32 keys), with k index 8g + j standing for key 32st + 16(j>>2) + 4g +
//   (j&3); the A operand V^T[d = 16db + i][that key] comes from the row-major V tile through two
//   transposing reads (ds_read_b64_tr_b16).  o[db][qb][r] is O[query 16qb + i][d = 16db + 4g + r].
//
// LDS image of a [rows][D] tile: 8-row x 64-byte subtiles,
//     off(row, c) = 16 D (row >> 3) + 512 (c >> 2) + 64 (row & 7) + 16 ((c & 3) ^ 2((row >> 2) & 1))
// (c = 16-byte chunk of the row).  Conflict-free for the K row reads (row = 16kb + i, chunk
// 4ks + g: the four rows of a ds_read_b128 lane group that share row & 3 get four different
// 16-byte columns) and for the transposing V reads (a 32-lane group reads rows 4g + (i >> 2),
// g in {0,1} or {2,3}: rows that differ by 4 get different 32-byte halves), and every fragment
// address is one of three per-lane bases plus an immediate.
#include "attn_mfma_common.h"
#include "tuning.h"

namespace myfa {
namespace {

constexpr int kWaves16 = 8;
constexpr int kThreads16 = kWaves16 * 64;
constexpr int kBQ16 = kWaves16 * 32;   // query rows per workgroup
constexpr int kBKV16 = 64;             // keys per tile
// A buffer offset no tensor reaches (the launcher keeps a head below it) and that a tile offset
// added to it cannot wrap: fetches from there read zero without touching memory.
constexpr int kPastEnd = 0x7FFF0000;

// SPEC = true: the running max is not looked for in every tile.  The weights of a tile are formed
// against the running max as it stands, and only when a lane's partial row sum passes `thr` (a
// limit `lim` = 2^thr: some weight then reached 2^(thr-4)) does the wave go back, find
// the tile's row maxima, move the running max, rescale O and l and form the weights again from the
// logits, which are still in their registers.  (The first tile and the masked tiles on the diagonal
// of a causal launch take the other path.)  16 v_max3, the exchange between the four lanes of
// a row and the vote leave the common path: a quarter of the vector phase's issue slots, and it is
// the vector issue port that bounds the loop.  The result is what the deferred max gives (weights
// up to the limit instead of 1, the same relative rounding); SPEC = false keeps the vote on the
// tile maxima (thr in log2 units; 0 = the textbook recurrence).
// STAMP builds are diagnostic only (tuning flavour): they bracket the phases with s_memtime and
// write the per-phase cycle sums of waves 0 and 4 to `dbg` (24 u64 per workgroup); never timed.
__device__ __forceinline__ unsigned long long stamp16() {
    __builtin_amdgcn_sched_barrier(0);
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
    __builtin_amdgcn_sched_barrier(0);
    return t;
}

// PRE (with SPEC): as in attn_fwd_stream_kernel -- Q is multiplied by scale * log2(e) once, when its
// fragments are read (one rounding to the element type per element), and the accumulators of
// S = K Q^T start from minus the running max, so the weights are exp2 of what the matrix cores
// deliver and the scale-and-subtract leaves the vector phase.  Launcher: only where the speculative
// softmax runs and no row statistic is asked for.
template <typename T, int D, bool CAUSAL = false, bool SPEC = true, bool STAMP = false, bool PRE = false>
__global__ void __launch_bounds__(kThreads16, D == 64 ? 4 : 2)
attn_fwd_mfma16_kernel(const T* __restrict__ Qg, const T* __restrict__ Kg, const T* __restrict__ Vg,
                       T* __restrict__ Og, float* __restrict__ LSEg, int N, int nqblk, int nwg,
                       float sl2, float thr, float lim, int gd, [[maybe_unused]] int probe,
                       [[maybe_unused]] unsigned long long* __restrict__ dbg) {
    [[maybe_unused]] unsigned long long t_k0 = 0, t_l0 = 0, t_l1 = 0, cyc[6] = {0, 0, 0, 0, 0, 0};
    if constexpr (STAMP) t_k0 = stamp16();
    using M = Mma16<T>;
    using vec8 = typename M::vec8;
    using vec4 = typename M::vec4;
    constexpr int CPR = D / 8;                   // 16-byte chunks per row
    constexpr int TILE_B = kBKV16 * D * 2;       // bytes per K (or V) tile
    constexpr int SUB_B = 16 * D;                // bytes of one 8-row group of the image
    constexpr int KS = D / 32;                   // k-steps of S^T = K Q^T
    constexpr int DB = D / 16;                   // 16-row blocks of O^T
    constexpr int LPT = kBKV16 * CPR / kThreads16;
    static_assert(LPT >= 1, "tile smaller than one load per thread");

    __shared__ __attribute__((aligned(16))) char lds[7 * TILE_B];   // K ring of 4, V ring of 3

    // Each XCD (blockIdx % 8 under round-robin dispatch) gets a contiguous range of
    // (head, query block) ids, so a head's K/V stay in one XCD's L2.  Speed only.
    int lid;
    {
        const int bid = blockIdx.x, xcd = bid & 7, idx = bid >> 3, qq = nwg >> 3, rr = nwg & 7;
        lid = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + idx;
    }
    const int head = lid / nqblk;
    // causal: query blocks differ in length; the long ones (high rows) of a head start first
    const int qblk = CAUSAL ? nqblk - 1 - (lid - head * nqblk) : lid - head * nqblk;
    const int q0 = qblk * kBQ16;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int grp = __builtin_amdgcn_readfirstlane(wave >> 2);   // 0: leads, 1: one phase behind
    const int i16 = lane & 15, g = lane >> 4;

    // gd <= D is the head dim in memory (a multiple of 8): rows are gd elements apart, and the
    // columns gd .. D-1 of the LDS images are zeros (their chunks are fetched from past the end of
    // the tensor, where the buffer bounds check answers zero), so Q K^T is unchanged and the
    // padded columns of O are never stored.
    const size_t hoff = (size_t)head * (size_t)N * gd;
    const int qrow0 = q0 + wave * 32 + i16;      // this lane's query of block 0; block 1: + 16

    // ---- K/V tiles go HBM/L2 -> LDS directly (buffer_load ... lds).  One wave-instruction fills
    // 1 KiB of LDS, lane L at base + 16 L, so the image is produced by choosing each lane's SOURCE
    // chunk: lane L of piece pc owns 16-byte unit u = 64 pc + L, which by the image formula is
    // (row, chunk) below.  Rows past N read as zero (buffer bounds check).
    const unsigned head_bytes = (unsigned)N * gd * 2;
    [[maybe_unused]] const int tile_gb = kBKV16 * gd * 2;          // bytes of one 64-row tile in memory
    const __amdgpu_buffer_rsrc_t krs = __builtin_amdgcn_make_buffer_rsrc((void*)(Kg + hoff), 0, head_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t vrs = __builtin_amdgcn_make_buffer_rsrc((void*)(Vg + hoff), 0, head_bytes, 0x00020000);
    [[maybe_unused]] const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    auto unit_src = [&](int u) __attribute__((always_inline)) -> int {   // byte offset in a [rows][D] tile
        const int grp8 = u / (SUB_B / 16), within = u % (SUB_B / 16);
        const int row = 8 * grp8 + ((within & 31) >> 2);
        const int ch = 4 * (within >> 5) + ((within & 3) ^ (2 * ((row >> 2) & 1)));
        return ch * 8 < gd ? (row * gd + ch * 8) * 2 : kPastEnd;
    };
    int st_goff[LPT];
#pragma unroll
    for (int i = 0; i < LPT; ++i) st_goff[i] = unit_src(64 * (wave * LPT + i) + lane);
    // Under the causal mask the tiles behind the workgroup's last one are asked for as the tile
    // past the end: zeros from the bounds check, no memory access, nothing for the epilogue to
    // wait for.
    [[maybe_unused]] const int t_past = (N + kBKV16 - 1) / kBKV16;
    [[maybe_unused]] const int t_stop = CAUSAL ? min(t_past, (q0 + kBQ16) / kBKV16) : t_past;
    auto fetch_piece = [&](const __amdgpu_buffer_rsrc_t& rs, int tile, int lds_byte, int i) __attribute__((always_inline)) {
#if defined(__HIP_DEVICE_COMPILE__)
        const int te = CAUSAL ? (tile < t_stop ? tile : t_past) : tile;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)(lds + lds_byte + 1024 * (wave_u * LPT + i)),
                                                 16, st_goff[i], te * tile_gb, 0, 0);
#endif
    };
    auto fetch_k = [&](int tile, int slot) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < LPT; ++i) fetch_piece(krs, tile, slot * TILE_B, i);
    };
    auto fetch_v = [&](int tile, int slot) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < LPT; ++i) fetch_piece(vrs, tile, (4 + slot) * TILE_B, i);
    };

    // ---- per-lane bases of the fragment reads (everything else is an immediate)
    // K (and Q) row read: row = 16 blk + i, chunk = 4 ks + g
    const int kbase = SUB_B * (i16 >> 3) + 64 * (i16 & 7) + 16 * (g ^ (2 * ((i16 >> 2) & 1)));
    // V transposing read: row = 32 st + 16 hi + 4 g + tq, chunk = 2 db + (tp >> 1), half tp & 1
    const int tq = i16 >> 2, tp = i16 & 3;
    int vbase[2];   // by db & 1
#pragma unroll
    for (int p = 0; p < 2; ++p)
        vbase[p] = SUB_B * (g >> 1) + 64 * (4 * (g & 1) + tq) + 16 * ((2 * p + (tp >> 1)) ^ (2 * (g & 1))) + 8 * (tp & 1);

    f32x4 o[DB][2];
#pragma unroll
    for (int db = 0; db < DB; ++db)
#pragma unroll
        for (int qb = 0; qb < 2; ++qb)
#pragma unroll
            for (int r = 0; r < 4; ++r) o[db][qb][r] = 0.f;
    float m[2] = {-INFINITY, -INFINITY}, l[2] = {0.f, 0.f};
    f32x4 s[4][2];
    vec8 qf[2][KS];
    vec8 pb[2][2];
    static_assert(!PRE || SPEC, "PRE goes with the speculative softmax");
    // PRE: what the accumulators of S start from, -c for the lane's two queries (c = the running max
    // they stand at; every register of a tuple holds the same value)
    f32x4 cneg[2];
#pragma unroll
    for (int qb = 0; qb < 2; ++qb)
#pragma unroll
        for (int r = 0; r < 4; ++r) cneg[qb][r] = 0.f;

    constexpr int NQ = 4 * KS, NP = 2 * DB, FW = D == 64 ? 4 : 8;   // (4, 6 and 12 at D = 128: 3-6 % more cycles)
    // fragment i of the matrix phase: i < NQ: K fragment (kb = i / KS, ks = i % KS) of the K slot;
    // otherwise V fragment (st, db) of the V slot.  Each fragment feeds two MFMAs (qb = 0, 1).
    // kaddr / vaddr[2]: per-lane base plus ring slot, formed once per phase (`slot_addr`); the
    // compiler must take them as opaque registers, or it folds the V ring's 64-KiB offset into
    // every read as an addition of its own (the offset field of a DS instruction ends at 65535).
    auto frag = [&](int kaddr, const int (&vaddr)[2], int i) __attribute__((always_inline)) -> s16x8 {
        if (i < NQ) {
            const int kb = i / KS, ks = i % KS;
            return *reinterpret_cast<const s16x8*>(lds + kaddr + (SUB_B * 2 * kb + 512 * ks));
        }
        const int j = i - NQ, st = j / DB, db = j % DB;
        const int imm = SUB_B * 4 * st + 512 * (db >> 1);
        const s16x4 lo = lds_read_tr(lds + vaddr[db & 1] + imm);
        const s16x4 hi = lds_read_tr(lds + vaddr[db & 1] + imm + SUB_B * 2);
        s16x8 a;
        a[0] = lo[0]; a[1] = lo[1]; a[2] = lo[2]; a[3] = lo[3];
        a[4] = hi[0]; a[5] = hi[1]; a[6] = hi[2]; a[7] = hi[3];
        return a;
    };
    // The matrix phase is one software-pipelined stream: the LDS fragment of step i + FW is
    // requested when the MFMAs of step i issue; the first FW requests (`prefill`) are made at the
    // end of the PREVIOUS vector phase, before the phase barrier.
    s16x8 win[FW];
    auto slot_addr = [&](int kslot, int vslot, int& kaddr, int (&vaddr)[2]) __attribute__((always_inline)) {
        kaddr = kbase + kslot * TILE_B;
        vaddr[0] = vbase[0] + (4 + vslot) * TILE_B;
        vaddr[1] = vbase[1] + (4 + vslot) * TILE_B;
        asm volatile("" : "+v"(kaddr), "+v"(vaddr[0]), "+v"(vaddr[1]));
    };
    auto prefill = [&](int kslot, int vslot, auto first_, auto last_) __attribute__((always_inline)) {
        constexpr int FIRST = decltype(first_)::value, LAST = decltype(last_)::value;
        int kaddr, vaddr[2];
        slot_addr(kslot, vslot, kaddr, vaddr);
#pragma unroll
        for (int i = FIRST; i < FIRST + FW; ++i)
            if (i < LAST) win[(i - FIRST) % FW] = frag(kaddr, vaddr, i);
    };
    auto mma_stream = [&](int kslot, int vslot, auto first_, auto last_, auto&& after) __attribute__((always_inline)) {
        constexpr int FIRST = decltype(first_)::value, LAST = decltype(last_)::value;
        int kaddr, vaddr[2];
        slot_addr(kslot, vslot, kaddr, vaddr);
        if constexpr (FIRST < NQ) {
#pragma unroll
            for (int kb = 0; kb < 4; ++kb)
#pragma unroll
                for (int qb = 0; qb < 2; ++qb) {
                    if constexpr (PRE) s[kb][qb] = cneg[qb];
                    else {
#pragma unroll
                        for (int r = 0; r < 4; ++r) s[kb][qb][r] = 0.f;
                    }
                }
        }
#pragma unroll
        for (int i = FIRST; i < LAST; ++i) {
            const s16x8 a = win[(i - FIRST) % FW];
            if (i < NQ) {
                s[i / KS][0] = M::run(a, qf[0][i % KS], s[i / KS][0]);
                s[i / KS][1] = M::run(a, qf[1][i % KS], s[i / KS][1]);
            } else {
                const int st = (i - NQ) / DB, db = (i - NQ) % DB;
                o[db][0] = M::run(a, pb[st][0], o[db][0]);
                o[db][1] = M::run(a, pb[st][1], o[db][1]);
            }
            if (i + FW < LAST) win[(i - FIRST) % FW] = frag(kaddr, vaddr, i + FW);
            after(i);
            // pin the order: left alone, the scheduler sinks every read to just before its MFMA
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    auto nothing = [](int) __attribute__((always_inline)) {};

    // MASK = false: a tile that lies before every row of the workgroup and inside N (causal builds
    // run those tiles in a loop of their own, whose code then is the code of the unmasked kernel).
    // The workgroup's first tile needs nothing special: against a running max of -inf every weight
    // and with it every row sum is infinite (or NaN where the logit is masked), which sends the
    // speculative path back and trips the other path's vote, and O and l are still zero when they
    // are scaled by exp2(-inf) = 0.  (A first tile with code of its own, no rescale in it, was
    // tried: the causal builds then spill 23-55 vector registers in their masked loop.)
    auto softmax = [&](int kv0, auto mask_flag, auto first_flag) __attribute__((always_inline)) {
        constexpr bool MASK = decltype(mask_flag)::value != 0;
        [[maybe_unused]] constexpr bool FIRST = decltype(first_flag)::value != 0;   // the workgroup's first tile
        // keys past N, and under the causal mask keys after the lane's query rows; a tile the
        // wave's first row already covers entirely needs no look
        if (MASK && (kv0 + kBKV16 > N || (CAUSAL && kv0 + kBKV16 - 1 > q0 + 32 * wave_u))) {
            // Key kv0 + 16 kb + 4 g + r is hidden from the lane's query of block qb when it lies past
            // lim = min(query row, N - 1): with t = lim - kv0 - 4 g per lane, when 16 kb + r > t.  Written
            // out as compare / select pairs on VCC with the tile-relative key as an inline constant: two
            // instructions per logit and no scalar registers (left to the compiler the 32 comparisons
            // come first, each mask in a scalar pair of its own, and the causal builds spill scalars).
            [[maybe_unused]] int t_[2];
#pragma unroll
            for (int qb = 0; qb < 2; ++qb) t_[qb] = (CAUSAL ? min(qrow0 + 16 * qb, N - 1) : N - 1) - kv0 - 4 * g;
            [[maybe_unused]] const float ninf = -INFINITY;
            static_for<4>([&](auto kb_) {
                static_for<4>([&](auto r_) {
                    constexpr int kb = decltype(kb_)::value, r = decltype(r_)::value;
#pragma unroll
                    for (int qb = 0; qb < 2; ++qb) s[kb][qb][r] = hide_past<16 * kb + r>(s[kb][qb][r], t_[qb], ninf);
                });
            });
        }
        // row max: four independent chains of v_max3_f32 (two per query), written out as single
        // instructions (vmax3 / vmax2: fmaxf would first quiet a possible signalling NaN in every
        // chain head, and one serial chain per query costs its latency 16 times over)
        auto lane_max = [&](float (&tmax)[2]) __attribute__((always_inline)) {
#pragma unroll
            for (int qb = 0; qb < 2; ++qb) {
                float a = vmax3(s[0][qb][0], s[0][qb][1], s[0][qb][2]);
                float b = vmax3(s[2][qb][0], s[2][qb][1], s[2][qb][2]);
                a = vmax3(a, s[0][qb][3], s[1][qb][0]);
                b = vmax3(b, s[2][qb][3], s[3][qb][0]);
                a = vmax3(a, s[1][qb][1], s[1][qb][2]);
                b = vmax3(b, s[3][qb][1], s[3][qb][2]);
                tmax[qb] = vmax3(a, b, vmax2(s[1][qb][3], s[3][qb][3]));
            }
        };
        // the running max moves to max(m, tile max); O and l follow
        auto move_max = [&](const float (&mt)[2]) __attribute__((always_inline)) {
#pragma unroll
            for (int qb = 0; qb < 2; ++qb) {
                const float m_new = fmaxf(m[qb], mt[qb]);
                const float alpha = __builtin_amdgcn_exp2f(m[qb] - m_new);
                m[qb] = m_new;
                l[qb] *= alpha;
#pragma unroll
                for (int db = 0; db < DB; ++db)
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[db][qb][r] *= alpha;
            }
        };
        // Only the tiles of a loop without mask code are speculated on: behind the mask code the
        // logits would have to stay alive beside the weights, and those few tiles (the diagonal
        // under the causal mask) then spill registers.
        constexpr bool SPECULATE = SPEC && !(CAUSAL && MASK);
        float psum[2];
        f32x4 p_[SPECULATE ? 4 : 1][2];
        auto& p = [&]() __attribute__((always_inline)) -> auto& { if constexpr (SPECULATE) return p_; else return s; }();
        // (Measured on this block: the same instructions in the compiler's order -- per query four
        // (scale-subtract, exp2) pairs, then (scale-subtract, add, exp2) triples -- run the kernel in
        // 7 % fewer cycles than either the two queries interleaved or any order pinned with
        // sched_group_barrier or written as one asm statement, which costs register copies.)
        auto weights = [&]() __attribute__((always_inline)) {
#pragma unroll
            for (int qb = 0; qb < 2; ++qb) {
#pragma unroll
                for (int kb = 0; kb < 4; ++kb)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        p[kb][qb][r] = __builtin_amdgcn_exp2f(__builtin_fmaf(s[kb][qb][r], sl2, -m[qb]));
                        if (kb == 0 && r == 0) psum[qb] = p[kb][qb][r];
                        else psum[qb] += p[kb][qb][r];
                    }
            }
        };
        if constexpr (PRE) {
            // the logits arrive as s' = scale log2(e) q.k - c
            auto weights_pre = [&](const float (&delta)[2], auto shifted) __attribute__((always_inline)) {
#pragma unroll
                for (int qb = 0; qb < 2; ++qb) {
#pragma unroll
                    for (int kb = 0; kb < 4; ++kb)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            if constexpr (decltype(shifted)::value) p[kb][qb][r] = __builtin_amdgcn_exp2f(s[kb][qb][r] - delta[qb]);
                            else p[kb][qb][r] = __builtin_amdgcn_exp2f(s[kb][qb][r]);
                            if (kb == 0 && r == 0) psum[qb] = p[kb][qb][r];
                            else psum[qb] += p[kb][qb][r];
                        }
                }
            };
            const float none[2] = {0.f, 0.f};
            // the first tile always looks for its maxima (m is -inf), and so do the masked tiles on the
            // diagonal of a causal launch (nothing is speculated on there)
            // (The first tile is told apart by a run-time flag the compiler cannot see through: with a
            // code path of its own for that one call the register allocation of the tile LOOP came out
            // with 100+ spills.  Its speculative weights are formed and thrown away.)
            bool back = true;
            if constexpr (SPECULATE) {
                int first_rt = FIRST;
                asm volatile("" : "+s"(first_rt));
                weights_pre(none, IC<0>{});
                // negated comparisons: a NaN or an infinite sum goes back too
                back = (__builtin_amdgcn_ballot_w64(!(psum[0] <= lim) || !(psum[1] <= lim)) != 0) | (first_rt != 0);   // wave-uniform
            }
            if (back) {
                float tmax[2];
                lane_max(tmax);
                const Pair mt_ = four_row_reduce(tmax[0], tmax[1], RawMaxOp{});
                const float mt[2] = {mt_.a - cneg[0][0], mt_.b - cneg[1][0]};   // the tile's maxima, shift undone
                float delta[2];
#pragma unroll
                for (int qb = 0; qb < 2; ++qb) delta[qb] = fmaxf(m[qb], mt[qb]) + cneg[qb][0];   // new max - c
                move_max(mt);
#pragma unroll
                for (int qb = 0; qb < 2; ++qb)
#pragma unroll
                    for (int r = 0; r < 4; ++r) cneg[qb][r] = -m[qb];
                weights_pre(delta, IC<1>{});
            }
        } else if constexpr (SPECULATE) {
            weights();
            // negated comparisons: a NaN or an infinite sum goes back too
            if (__builtin_amdgcn_ballot_w64(!(psum[0] <= lim) || !(psum[1] <= lim)) != 0) {   // wave-uniform
                float tmax[2];
                lane_max(tmax);
                const Pair mt_ = four_row_reduce(tmax[0], tmax[1], RawMaxOp{});
                const float mt[2] = {mt_.a * sl2, mt_.b * sl2};
                move_max(mt);
                weights();
            }
        } else {
            float tmax[2];
            lane_max(tmax);
            // A row's max passes the threshold exactly when one of its four lanes' partial maxima does,
            // so the lanes can vote on their own 16 logits and leave the exchange between the four lanes
            // of a row to the (rare) tile in which the running max has to move: 12 instructions fewer per
            // tile.  Measured: at D = 64, where the vector issue port is the bound, that is what it saves;
            // at D = 128 the kernel takes 3.6 % MORE cycles with it (the compiler's schedule of the phase
            // changes), so there the exchange stays in front of the vote.
            constexpr bool LAZY = D == 64;
            Pair mt_ = {tmax[0], tmax[1]};
            if constexpr (!LAZY) mt_ = four_row_reduce(tmax[0], tmax[1], RawMaxOp{});
            float mt[2] = {mt_.a * sl2, mt_.b * sl2};
            if (__builtin_amdgcn_ballot_w64((mt[0] - m[0] > thr) || (mt[1] - m[1] > thr)) != 0) {   // wave-uniform
                if constexpr (LAZY) {
                    mt_ = four_row_reduce(tmax[0], tmax[1], RawMaxOp{});
                    mt[0] = mt_.a * sl2;
                    mt[1] = mt_.b * sl2;
                }
                move_max(mt);
            }
            weights();
        }
        l[0] += psum[0];
        l[1] += psum[1];
#pragma unroll
        for (int st = 0; st < 2; ++st)
#pragma unroll
            for (int qb = 0; qb < 2; ++qb)
#pragma unroll
                for (int j = 0; j < 8; ++j) pb[st][qb][j] = (T)p[2 * st + (j >> 2)][qb][j & 3];
        // The conversions belong to the vector phase: left alone, the compiler sinks them (and the two
        // additions to l) below the phase barrier, in front of the matrix phase's first MFMA -- 80
        // cycles at the head of the phase that is the critical path.  Pinned here: +3.3 % (D = 128).
#pragma unroll
        for (int st = 0; st < 2; ++st)
#pragma unroll
            for (int qb = 0; qb < 2; ++qb) {
                u32x4 w = __builtin_bit_cast(u32x4, pb[st][qb]);
                asm volatile("" : "+v"(w));
                pb[st][qb] = __builtin_bit_cast(vec8, w);
            }
        asm volatile("" : "+v"(l[0]), "+v"(l[1]));
    };

    // key tiles of this workgroup (causal: up to the block's last row; the first tile always
    // holds key 0, which every row may see, so no row is ever fully masked there)
    const int T_ = CAUSAL ? min((N + kBKV16 - 1) / kBKV16, (q0 + kBQ16) / kBKV16) : (N + kBKV16 - 1) / kBKV16;

    auto dma_wait = [&]() __attribute__((always_inline)) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); };
    auto phase_barrier = [&]() __attribute__((always_inline)) {
        asm volatile("" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };

    // ---- prologue: K(0), V(0), K(1) into their ring slots, and this wave's 32 x D block of Q
    // into a private half slot (K slots 2, 3 and V slots 1, 2 are free until the first barrier),
    // laid out like the first 32 rows of a tile.
    fetch_k(0, 0);
    fetch_v(0, 0);
    fetch_k(1, 1);
    const int qbyte = (wave_u < 2 ? 2 : wave_u < 4 ? 3 : wave_u < 6 ? 5 : 6) * TILE_B + (wave_u & 1) * (TILE_B / 2);
    {
        [[maybe_unused]] const __amdgpu_buffer_rsrc_t qrs = __builtin_amdgcn_make_buffer_rsrc((void*)(Qg + hoff), 0, head_bytes, 0x00020000);
        [[maybe_unused]] const int qrow0_b = (q0 + 32 * wave) * gd * 2;   // in the lane offset: it is bounds-checked
#pragma unroll
        for (int pc = 0; pc < TILE_B / 2 / 1024; ++pc) {
            [[maybe_unused]] const int src = unit_src(64 * pc + lane);
#if defined(__HIP_DEVICE_COMPILE__)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(qrs, (lds_void*)(lds + qbyte + 1024 * pc), 16,
                                                     src == kPastEnd ? kPastEnd : qrow0_b + src, 0, 0, 0);
#endif
        }
    }
    dma_wait();
    // Q fragments out of the private block; the reads must have returned before the barrier,
    // behind which K(2) and V(1) overwrite the Q blocks.
#pragma unroll
    for (int qb = 0; qb < 2; ++qb)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
            qf[qb][ks] = *reinterpret_cast<const vec8*>(lds + (kbase + qbyte) + (SUB_B * 2 * qb + 512 * ks));
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if constexpr (PRE) {
#pragma unroll
        for (int qb = 0; qb < 2; ++qb)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                for (int j = 0; j < 8; ++j) qf[qb][ks][j] = (T)((float)qf[qb][ks][j] * sl2);
    }
    __syncthreads();

    prefill(0, 0, IC<0>{}, IC<NQ>{});
    // S(0) = K(0) Q^T, with the fetches of K(2), V(1) behind its last fragment request
    mma_stream(0, 0, IC<0>{}, IC<NQ>{}, [&](int i) __attribute__((always_inline)) {
        constexpr int L0 = NQ - FW;
#pragma unroll
        for (int j = 0; j < LPT; ++j) {
            if (i == L0 + 2 * j)     fetch_piece(krs, 2, 2 * TILE_B, j);
            if (i == L0 + 2 * j + 1) fetch_piece(vrs, 1, (4 + 1) * TILE_B, j);
        }
    });
    if (grp) __syncthreads();
    softmax(0, IC<1>{}, IC<1>{});
    dma_wait();                              // K(2), V(1): every wave's share is in before the barrier
    prefill(1, 0, IC<0>{}, IC<NQ + NP>{});   // first fragments of M(0): K(1), complete since the prologue
    phase_barrier();

    // Ring bookkeeping (wave-uniform): tile u of K lives in slot u % 4, of V in slot u % 3.  The
    // fetches of K(t+3) and V(t+2) are issued under the last MFMAs of M(t), behind the phase's last
    // fragment request, and every wave waits for its own at the end of its X(t+1).
    int k1 = 1, v0 = 0;              // (t+1) % 4, t % 3
    auto tile_step = [&](int t, auto mask_flag) __attribute__((always_inline)) {
        [[maybe_unused]] unsigned long long ta = 0, tb = 0, tc = 0, tx = 0, td = 0, te = 0;
        if constexpr (STAMP) ta = stamp16();
        const int k2 = (k1 + 1) & 3;             // (t+2) % 4
        const int k3 = (k1 + 2) & 3;             // (t+3) % 4
        const int v1 = v0 == 2 ? 0 : v0 + 1;     // (t+1) % 3
        const int v2 = v1 == 2 ? 0 : v1 + 1;     // (t+2) % 3
        // ---- M(t): S(t+1) = K(t+1) Q^T ; O += V(t)^T P(t)^T ; fetch K(t+3), V(t+2)
        mma_stream(k1, v0, IC<0>{}, IC<NQ + NP>{}, [&](int i) __attribute__((always_inline)) {
            constexpr int L0 = NQ + NP - FW;     // no fragment requests from here on
#pragma unroll
            for (int j = 0; j < LPT; ++j) {
                if constexpr (STAMP) { if (probe & 8) continue; }   // timing probe: no fetches (wrong results)
                if (i == L0 + 2 * j)     fetch_piece(krs, t + 3, k3 * TILE_B, j);
                if (i == L0 + 2 * j + 1) fetch_piece(vrs, t + 2, (4 + v2) * TILE_B, j);
            }
        });
        if constexpr (STAMP) tb = stamp16();
        phase_barrier();                         // the fetches just issued stay in flight across it
        if constexpr (STAMP) tc = stamp16();
        // ---- X(t+1)
        if constexpr (STAMP) {   // timing probe: no vector phase (wrong results)
            if (!(probe & 16)) softmax((t + 1) * kBKV16, mask_flag, IC<0>{});
        } else {
            softmax((t + 1) * kBKV16, mask_flag, IC<0>{});
        }
        if constexpr (STAMP) tx = stamp16();
        dma_wait();                              // this wave's share of K(t+3), V(t+2) is in LDS
        prefill(k2, v1, IC<0>{}, IC<NQ + NP>{}); // first fragments of M(t+1): K(t+2)
        if constexpr (STAMP) td = stamp16();
        phase_barrier();
        if constexpr (STAMP) {
            te = stamp16();
            cyc[0] += tb - ta; cyc[1] += tc - tb; cyc[2] += tx - tc; cyc[3] += td - tx; cyc[4] += te - td;
        }
        k1 = k2; v0 = v1;
    };
    int t = 0;
    if constexpr (STAMP) t_l0 = stamp16();
    if constexpr (CAUSAL) {
        // tiles 1 .. t_clean-1 lie before the workgroup's first row and inside N: no mask code
        const int t_clean = min(q0, N) / kBKV16;
        for (; t + 1 < min(T_, t_clean); ++t) tile_step(t, IC<0>{});
    }
    for (; t + 1 < T_; ++t) tile_step(t, IC<1>{});
    if constexpr (STAMP) t_l1 = stamp16();
    prefill(0, v0, IC<NQ>{}, IC<NQ + NP>{});
    mma_stream(0, v0, IC<NQ>{}, IC<NQ + NP>{}, nothing);   // M(T-1): O += V(T-1)^T P^T
    dma_wait();
    if (grp == 0) __syncthreads();   // the leading half meets the lagging half's last barrier

    const Pair ltot = four_row_reduce(l[0], l[1], SumOp{});
    const float inv[2] = {1.0f / ltot.a, 1.0f / ltot.b};
    {
        // O leaves through LDS so that it is stored as whole rows: each wave parks its 32 x D tile
        // in its own region (rows padded by 16 bytes), then 64/CPR rows go out per store
        // instruction, 16 bytes per lane, each row contiguous.
        constexpr int OROW_B = D * 2 + 16;
        __syncthreads();                         // every wave is done with the K/V rings
        char* obuf = lds + wave * (32 * OROW_B);
#pragma unroll
        for (int qb = 0; qb < 2; ++qb)
#pragma unroll
            for (int db = 0; db < DB; ++db) {
                vec4 w;
#pragma unroll
                for (int r = 0; r < 4; ++r) w[r] = (T)(o[db][qb][r] * inv[qb]);
                *reinterpret_cast<vec4*>(obuf + (16 * qb + i16) * OROW_B + (16 * db + 4 * g) * 2) = w;
            }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // own writes only: the region is per wave
        constexpr int RPI = 64 / CPR;            // rows per store instruction
        const int orow0 = q0 + wave * 32;
#pragma unroll
        for (int k2 = 0; k2 < 32 / RPI; ++k2) {
            const int row = k2 * RPI + lane / CPR, ch = lane % CPR;
            const u32x4 val = *reinterpret_cast<const u32x4*>(obuf + row * OROW_B + ch * 16);
            if (orow0 + row < N && ch * 8 < gd)
                *reinterpret_cast<u32x4*>(Og + hoff + (size_t)(orow0 + row) * gd + ch * 8) = val;
        }
        // lanes 0-15 hold block 0's statistics for their query, lanes 16-31 write block 1's
        if (LSEg != nullptr && g < 2) {
            const int qr = qrow0 + 16 * g;
            const float mm = g == 0 ? m[0] : m[1], ll = g == 0 ? ltot.a : ltot.b;
            if (qr < N) LSEg[(size_t)head * N + qr] = (mm + __builtin_amdgcn_logf(ll)) * 0.6931471805599453f;
        }
    }
    if constexpr (STAMP) {
        if (dbg != nullptr && (wave == 0 || wave == 4) && lane == 0) {
            unsigned long long* o_ = dbg + (size_t)blockIdx.x * 24 + (wave >> 2) * 12;
#pragma unroll
            for (int i = 0; i < 5; ++i) o_[i] = cyc[i];
            o_[5] = t_l0 - t_k0;
            o_[6] = stamp16() - t_l1;
            o_[7] = T_;
        }
    }
}

template <typename T, int D>
int launch16(const void* q, const void* k, const void* v, void* o, float* lse,
             int64_t heads, int64_t N, int64_t d, float scale, float thr, bool spec, bool pre, int probe, bool causal, hipStream_t s) {
    if (N * d * 2 >= (int64_t)kPastEnd) return fail(MYFA_E_UNSUPPORTED, "attention: N*d=%lld per head too large", (long long)(N * d));
    const int64_t nqblk = cdiv(N, kBQ16);
    const int64_t nwg = heads * nqblk;
    if (nwg > 0x7fffffffLL) return fail(MYFA_E_UNSUPPORTED, "attention: too many workgroups (%lld)", (long long)nwg);
    const float sl2 = scale * 1.4426950408889634f;
    const float lim = exp2f(thr);   // the speculative path compares row sums, not logits
    [[maybe_unused]] unsigned long long* dbg = tuning().attn_dbg;
    auto go = [&](auto kern, unsigned long long* dbg_) {
        launch_kernel(kern, dim3((unsigned)nwg), dim3(kThreads16), 0, s,
                           static_cast<const T*>(q), static_cast<const T*>(k), static_cast<const T*>(v),
                           static_cast<T*>(o), lse, (int)N, (int)nqblk, (int)nwg, sl2, thr, lim, (int)d, probe, dbg_);
    };
#ifdef MYFA_TUNING
    if (dbg != nullptr && D == 128 && !causal) {
        if (spec) go(attn_fwd_mfma16_kernel<T, 128, false, true, true>, dbg);
        else go(attn_fwd_mfma16_kernel<T, 128, false, false, true>, dbg);
        return check_launch("attn_fwd_mfma16 (stamped)");
    }
#endif
    // Under the causal mask the speculative form loses (941 against 1004 TFLOP/s at the headline shape,
    // 726 against 777 at d = 64: the masked tiles take the other path anyway and the build carries both).
    if (causal && spec && pre) go(attn_fwd_mfma16_kernel<T, D, true, true, false, true>, nullptr);
    else if (causal) go(attn_fwd_mfma16_kernel<T, D, true, false>, nullptr);
    else if (spec && pre) go(attn_fwd_mfma16_kernel<T, D, false, true, false, true>, nullptr);
    else if (spec) go(attn_fwd_mfma16_kernel<T, D, false, true>, nullptr);
    else go(attn_fwd_mfma16_kernel<T, D, false, false>, nullptr);
    return check_launch("attn_fwd_mfma16");
}

}  // namespace

// dtype: MYFA_F16 or MYFA_BF16; d: a multiple of 8 in 8..128; heads = B*H.
//
// spec: 1 = the speculative build, 0 = the build that votes on the tile maxima, < 0 = the library's
// choice, which goes by how peaked the softmax of unit-variance inputs would be: their logits have a
// standard deviation of scale * log2(e) * sqrt(d) in base-2 units -- 1.44 at scale = d^-1/2, 16.3
// with the reference's unscaled logits at d = 128 (tests/test_flash_attn.py:165).  Up to 4 the
// speculative build runs (+3..4 % at d = 128, +7..10 % at d = 64; rows are sums of many comparable
// weights, and a stale max costs no accuracy: max |error| 1.6e-3 in bfloat16 at any limit).  Beyond,
// a row is usually one dominant key, the running max has to move in a third of the tiles, and what
// a stale max costs there is accuracy: the dominant key's weight is then some 2^x instead of the
// exactly representable 1, its rounding (relative 2^-8 in bfloat16) no longer cancels against the
// float32 row sum, and |error| reaches 0.031 (4 elements in 12.6 million above 2e-2 with the vote
// at thr = 8, 15 with a limit of 2^30; none at thr = 0, where the bound is the output's own
// rounding, 0.0156 for |O| in [4, 8)).  There the vote at thr = 8 runs, as before.
// thr < 0: the library's choice: 8 for the vote; for the speculative build a lane's partial row sum
// may reach 2^12 with float16 weights (largest finite value 65504) and 2^30 with bfloat16 weights,
// which have float32's exponent range.
int attn_fwd_mfma16(const void* q, const void* k, const void* v, void* o, float* lse,
                    int64_t heads, int64_t N, int64_t d, float scale, float thr, int spec, int pre, int probe,
                    int dtype, bool causal, hipStream_t s) {
    const bool sp = spec < 0 ? scale * 1.4426950408889634f * sqrtf((float)d) <= 4.0f : spec != 0;
    // pre: < 0 = the library's choice (never, in this build), else forced on / off (tuning flavour)
    const bool pr = sp && pre > 0;
    if (thr < 0.f) thr = !sp || (causal && !pr) ? 8.f : dtype == MYFA_BF16 ? 30.f : 12.f;
    if (dtype == MYFA_F16 && thr > 15.f) thr = 15.f;
    if (thr > 60.f) thr = 60.f;
    // head dims that are multiples of 8 run on the next build up, the missing columns read as zeros
    if (d >= 8 && d <= 128 && d % 8 == 0) {
        const bool big = d > 64;
        if (dtype == MYFA_BF16) return big ? launch16<__bf16, 128>(q, k, v, o, lse, heads, N, d, scale, thr, sp, pr, probe, causal, s)
                                           : launch16<__bf16, 64>(q, k, v, o, lse, heads, N, d, scale, thr, sp, pr, probe, causal, s);
        if (dtype == MYFA_F16)  return big ? launch16<_Float16, 128>(q, k, v, o, lse, heads, N, d, scale, thr, sp, pr, probe, causal, s)
                                           : launch16<_Float16, 64>(q, k, v, o, lse, heads, N, d, scale, thr, sp, pr, probe, causal, s);
    }
    return fail(MYFA_E_UNSUPPORTED, "attention (f16/bf16): head dim %lld is not a multiple of 8 in 8..128", (long long)d);
}

}  // namespace myfa
