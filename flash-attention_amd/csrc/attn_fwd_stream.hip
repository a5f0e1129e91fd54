// Fused attention forward for f16 / bf16 on gfx950 matrix cores: the resident-workgroup build.
//
//     O[b,h] = softmax(scale * Q[b,h] K[b,h]^T) V[b,h]      Q,K,V,O: [B,H,N,D], D in {64,128}
//
// The kernel the reference sketches and leaves as a TODO
// (csrc/my_flash_attn/flash_attention.cu:23-69); the math is the online-softmax recurrence of
// its Python restatement (tests/test_flash_attn.py:82-93).
//
// Register layouts, LDS image, LDS-DMA rings and the two-phase schedule are those of
// attn_fwd_mfma16_kernel (attn_fwd_mfma16.hip, whose header describes them).  What differs is
// the life of a workgroup.  There a workgroup computes ONE block of 256 query rows: it fills its
// rings (K(0), V(0), K(1), Q: a burst of 112 KiB that every workgroup on the chip issues at the
// same moment), runs its tiles, drains the rings and stores O -- and only one workgroup fits a CU
// at D = 128, so nothing runs on the CU's matrix cores during the first and the last ~6 % of that
// life.  Here the grid is as many workgroups as the chip holds at once, each working through a
// LIST of (head, query block) items as one unbroken stream of tiles:
//
//   * the K and V rings never drain: the fetches that run 3 (K) and 2 (V) tiles ahead of the
//     matrix phase simply move on to the next item's first tiles (`Cursor`);
//   * the next item's Q fragments are loaded straight into their registers (buffer loads, no LDS
//     staging) right behind the matrix phase that used the old ones for the last time, and have a
//     whole vector phase to arrive;
//   * the matrix phase of an item's last tile already computes S(0) of the next item;
//   * between that phase and the next item's first vector phase the wave normalises and stores its
//     32 x D block of O (through a small LDS region of its own, 16 rows at a time, so the rings
//     stay live) and clears O, m and l.
// An item boundary so costs the matrix cores the time of two such stores (one per wave half)
// instead of a drain and a refill.
//
// Items are dealt so that each XCD (blockIdx % 8 under round-robin dispatch) works on a contiguous
// range of (head, query block) ids and its workgroups are, at any moment, on neighbouring query
// blocks of the same few heads (K/V reuse in that XCD's L2).  Under the causal mask the unit that
// is dealt is a pair of query blocks of one head that mirror each other (block nqblk-1-p, then
// block p), so every unit has the same number of tiles.  Speed only.
#include "attn_mfma_common.h"
#include "tuning.h"

namespace myfa {
namespace {

constexpr int kWavesS = 8;
constexpr int kThreadsS = kWavesS * 64;
constexpr int kBQS = kWavesS * 32;     // query rows per item
constexpr int kBKVS = 64;              // keys per tile
// A buffer offset no tensor reaches (the launcher keeps a head below it) and that a tile offset
// added to it cannot wrap: fetches from there read zero without touching memory.
constexpr int kPastEndS = 0x7FFF0000;

// SPEC as in attn_fwd_mfma16_kernel: true = the weights of a tile are formed against the running
// max as it stands and the tile's maxima are looked for only when a partial row sum passes `lim`;
// false = the max of every tile, vote on its growth beyond `thr`.
// PRE (with SPEC, without the mask): the logits leave the matrix cores ready for the exponential.
// Q is multiplied by scale * log2(e) when it arrives (one rounding to the element type per element,
// once per item) and the accumulators of S = K Q^T start from minus the running max instead of
// zero, so S' = scale log2(e) Q K^T - c comes out of the MFMA chain and the weights are exp2(S'):
// the scale-and-subtract, a third of the vector phase's instructions, is gone from the common path.
// `c` is the max the accumulators started from: the running max, except in an item's first tile,
// whose S was formed beside the previous item's last tile and carries that item's max.
// The launcher uses it only where logits are small (the regime it picks SPEC in): rounding
// Q * scale * log2(e) to bfloat16 moves a logit by ~2^-9 of its size, 0.005 base-2 units at most there.
// WIDE (D = 64 only): one workgroup per CU with 256 registers per lane instead of two with 128.
// LM (with SPEC, without PRE and the mask): the row sums come out of the matrix cores -- one more MFMA
// per 32 keys and query block, a fragment of ones against the weights' B operand -- so they are sums
// of the ROUNDED weights: numerator and denominator of O carry the same rounding, which is what
// lets the speculative softmax run on large (unscaled) logits, where a row is mostly one dominant
// key and its weight's rounding would otherwise stand against an exact sum.  The go-back test looks
// at the largest weight instead of the row sum.
template <typename T, int D, bool CAUSAL, bool SPEC, bool PRE = false, bool WIDE = false, bool LM = false>
__global__ void __launch_bounds__(kThreadsS, D == 64 && !WIDE ? 4 : 2)
attn_fwd_stream_kernel(const T* __restrict__ Qg, const T* __restrict__ Kg, const T* __restrict__ Vg,
                       T* __restrict__ Og, float* __restrict__ LSEg, int N, int nqblk, int hp, int nunits,
                       float sl2, float thr, float lim, int gd) {
    using M = Mma16<T>;
    using vec8 = typename M::vec8;
    using vec4 = typename M::vec4;
    constexpr int CPR = D / 8;                   // 16-byte chunks per row
    constexpr int TILE_B = kBKVS * D * 2;        // bytes per K (or V) tile
    constexpr int SUB_B = 16 * D;                // bytes of one 8-row group of the image
    constexpr int KS = D / 32;                   // k-steps of S^T = K Q^T
    constexpr int DB = D / 16;                   // 16-row blocks of O^T
    constexpr int LPT = kBKVS * CPR / kThreadsS;
    static_assert(LPT >= 1, "tile smaller than one load per thread");
    constexpr int OROW_B = D * 2 + 16;           // a parked row of O, padded by 16 bytes
    constexpr int PARK_B = 16 * OROW_B;          // per wave: 16 rows at a time
    constexpr int RING_B = 7 * TILE_B;           // K ring of 4, V ring of 3

    __shared__ __attribute__((aligned(16))) char lds[RING_B + kWavesS * PARK_B];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave_u = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave_u >> 2;                 // 0: leads, 1: one phase behind
    const int i16 = lane & 15, g = lane >> 4;

    // ---- this workgroup's list.  A UNIT is one query block, or under the causal mask a pair of
    // query blocks of one head that mirror each other (block nqblk-1-p, then block p; the middle
    // block of an odd count stands alone), so that every unit has the same number of tiles and a
    // static split of the units is a balanced one.  `hp` = units per head.  Workgroups b, b + 8, ...
    // share an XCD (a grid of fewer than 8 workgroups has that many groups of one); group x owns the
    // contiguous unit ids [lo, lo + cnt) and workgroup j of its nj takes lo + j, lo + nj + j, ...
    int lo, nj, jj, my_units;
    {
        const int G = gridDim.x, bid = blockIdx.x, nx = G < 8 ? G : 8;
        const int xcd = bid % nx, qq = nunits / nx, rr = nunits % nx;
        jj = bid / nx;
        lo = xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq;
        const int cnt = qq + (xcd < rr ? 1 : 0);
        nj = (G - xcd + nx - 1) / nx;
        my_units = cnt > jj ? (cnt - jj + nj - 1) / nj : 0;
    }
    if (my_units == 0) return;                   // (whole workgroup; the launcher sizes the grid so that it does not happen)
    const int t_all = (N + kBKVS - 1) / kBKVS;   // key tiles of a head
    // An ITEM is one query block: 256 rows of one head against its nt key tiles.  Seq = a position
    // in the workgroup's list of items: unit ps of the workgroup, and which block of the pair.
    struct Seq { int ps, half; };
    struct Item { int head, q0, nt, pair2; };    // pair2: the unit's second block follows this one
    auto decode = [&](Seq q) __attribute__((always_inline)) -> Item {
        const int uid = lo + q.ps * nj + jj;
        const int head = uid / hp;
        const int p = uid - head * hp;
        const int qblk = CAUSAL ? (q.half ? p : nqblk - 1 - p) : p;
        const int pair2 = CAUSAL && q.half == 0 && 2 * p != nqblk - 1;
        const int q0 = qblk * kBQS;
        // causal: tiles up to the block's last row (the first tile always holds key 0, which every
        // row may see, so no row is ever fully masked there)
        const int T_ = CAUSAL ? min(t_all, (q0 + kBQS) / kBKVS) : t_all;
        return {__builtin_amdgcn_readfirstlane(head), __builtin_amdgcn_readfirstlane(q0), __builtin_amdgcn_readfirstlane(T_),
                __builtin_amdgcn_readfirstlane(pair2)};
    };
    auto seq_next = [&](Seq q, int pair2) __attribute__((always_inline)) -> Seq {
        return pair2 ? Seq{q.ps, 1} : Seq{q.ps + 1, 0};
    };

    // gd <= D is the head dim in memory (a multiple of 8): rows are gd elements apart, and the
    // columns gd .. D-1 of the LDS images and of the Q fragments are zeros (fetched from past the
    // end of the tensor, where the buffer bounds check answers zero), so Q K^T is unchanged and
    // the padded columns of O are never stored.
    const unsigned head_bytes = (unsigned)N * gd * 2;
    const size_t head_elems = (size_t)N * gd;
    [[maybe_unused]] const int tile_gb = kBKVS * gd * 2;          // bytes of one 64-row tile in memory

    // ---- K/V tiles go HBM/L2 -> LDS directly (buffer_load ... lds).  One wave-instruction fills
    // 1 KiB of LDS, lane L at base + 16 L, so the image is produced by choosing each lane's SOURCE
    // chunk: lane L of piece pc owns 16-byte unit u = 64 pc + L, which by the image formula is
    // (row, chunk) below.  Rows past N read as zero (buffer bounds check).
    auto unit_src = [&](int u) __attribute__((always_inline)) -> int {   // byte offset in a [rows][D] tile
        const int grp8 = u / (SUB_B / 16), within = u % (SUB_B / 16);
        const int row = 8 * grp8 + ((within & 31) >> 2);
        const int ch = 4 * (within >> 5) + ((within & 3) ^ (2 * ((row >> 2) & 1)));
        return ch * 8 < gd ? (row * gd + ch * 8) * 2 : kPastEndS;
    };
    int st_goff[LPT];
#pragma unroll
    for (int i = 0; i < LPT; ++i) st_goff[i] = unit_src(64 * (wave_u * LPT + i) + lane);

    // A fetch cursor walks the tiles of the workgroup's items in stream order, ahead of the
    // matrix phases.  Behind the last item it stays past the end of a head (zeros from the bounds
    // check, no memory access), so the stream needs no branch around its fetches.
    struct Cursor { Seq q; int tile, nt, pair2, boff; const T* base; };
    auto cursor_at = [&](Seq q, const T* tensor) __attribute__((always_inline)) -> Cursor {
        if (q.ps < my_units) {
            const Item it = decode(q);
            return {q, 0, it.nt, it.pair2, 0, tensor + (size_t)it.head * head_elems};
        }
        return {q, 0, 0x7fffffff, 0, t_all * tile_gb, tensor};
    };
    auto advance = [&](Cursor& c, const T* tensor) __attribute__((always_inline)) {
        c.boff += tile_gb;
        if (++c.tile == c.nt) c = cursor_at(seq_next(c.q, c.pair2), tensor);
    };
    auto fetch_piece = [&](const Cursor& c, int lds_byte, int i) __attribute__((always_inline)) {
#if defined(__HIP_DEVICE_COMPILE__)
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)c.base, 0, head_bytes, 0x00020000);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)(lds + lds_byte + 1024 * (wave_u * LPT + i)),
                                                 16, st_goff[i], c.boff, 0, 0);
#endif
    };
    Cursor kc = cursor_at(Seq{0, 0}, Kg), vc = cursor_at(Seq{0, 0}, Vg);

    // ---- per-lane bases of the fragment reads (everything else is an immediate)
    // K row read: row = 16 blk + i, chunk = 4 ks + g
    const int kbase = SUB_B * (i16 >> 3) + 64 * (i16 & 7) + 16 * (g ^ (2 * ((i16 >> 2) & 1)));
    // V transposing read: row = 32 st + 16 hi + 4 g + tq, chunk = 2 db + (tp >> 1), half tp & 1
    const int tq = i16 >> 2, tp = i16 & 3;
    int vbase[2];   // by db & 1
#pragma unroll
    for (int p = 0; p < 2; ++p)
        vbase[p] = SUB_B * (g >> 1) + 64 * (4 * (g & 1) + tq) + 16 * ((2 * p + (tp >> 1)) ^ (2 * (g & 1))) + 8 * (tp & 1);

    static_assert(!LM || (SPEC && !PRE && !CAUSAL), "LM goes with the unmasked speculative softmax on unscaled Q");
    f32x4 o[DB][2];
    float m[2], l[2];
    [[maybe_unused]] f32x4 lacc[2];              // LM: the row sums, every register of a tuple the lane's query's
    [[maybe_unused]] s16x8 ones;                 // LM: an A fragment of ones
    if constexpr (LM) {
        vec8 one8;
#pragma unroll
        for (int j = 0; j < 8; ++j) one8[j] = (T)1.0f;
        u32x4 w = __builtin_bit_cast(u32x4, one8);
        asm volatile("" : "+v"(w));              // (kept in registers: an MFMA operand cannot be a constant)
        ones = __builtin_bit_cast(s16x8, w);
    }
    auto clear_state = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int db = 0; db < DB; ++db)
#pragma unroll
            for (int qb = 0; qb < 2; ++qb)
#pragma unroll
                for (int r = 0; r < 4; ++r) o[db][qb][r] = 0.f;
        m[0] = m[1] = -INFINITY;
        l[0] = l[1] = 0.f;
        if constexpr (LM) {
#pragma unroll
            for (int qb = 0; qb < 2; ++qb)
#pragma unroll
                for (int r = 0; r < 4; ++r) lacc[qb][r] = 0.f;
        }
    };
    clear_state();
    f32x4 s[4][2];
    vec8 qf[2][KS];
    vec8 pb[2][2];
    static_assert(!PRE || SPEC, "PRE goes with the speculative softmax");
    // PRE: what the accumulators of S start from, -c for the lane's two queries (every register of a
    // tuple holds the same value: a lane's four accumulator rows are keys of one query)
    f32x4 cneg[2];
#pragma unroll
    for (int qb = 0; qb < 2; ++qb)
#pragma unroll
        for (int r = 0; r < 4; ++r) cneg[qb][r] = 0.f;
    [[maybe_unused]] int q_fresh = 0;            // PRE: Q fragments have arrived unscaled (wave-uniform)
    auto scale_q = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int qb = 0; qb < 2; ++qb)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                for (int j = 0; j < 8; ++j) qf[qb][ks][j] = (T)((float)qf[qb][ks][j] * sl2);
    };

    // Q fragments of an item, straight into their registers: lane (i, g) holds, for query block qb
    // and k-step ks, Q[q0 + 32 wave + 16 qb + i][32 ks + 8 g .. + 7].  Rows past N and columns past
    // gd read as zero.  (Compiler-visible loads, so that the compiler places the waits and never
    // spills a fragment whose load is still in flight.)
    auto load_q = [&](const Item& it) __attribute__((always_inline)) {
        [[maybe_unused]] const __amdgpu_buffer_rsrc_t qrs =
            __builtin_amdgcn_make_buffer_rsrc((void*)(Qg + (size_t)it.head * head_elems), 0, head_bytes, 0x00020000);
        const int row_b = (it.q0 + 32 * wave_u) * gd * 2;     // (in the lane offset: it is bounds-checked)
        const int q_lane = (i16 * gd + 8 * g) * 2;
#pragma unroll
        for (int qb = 0; qb < 2; ++qb)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                [[maybe_unused]] const int off = (4 * ks + g) * 8 < gd ? q_lane + row_b + 32 * qb * gd + 64 * ks : kPastEndS;
#if defined(__HIP_DEVICE_COMPILE__)
                qf[qb][ks] = __builtin_bit_cast(vec8, __builtin_amdgcn_raw_buffer_load_b128(qrs, off, 0, 0));
#endif
            }
    };

    constexpr int NQ = 4 * KS, NP = 2 * DB, FW = D == 64 ? 4 : 8;
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
        const s16x4 lo4 = lds_read_tr(lds + vaddr[db & 1] + imm);
        const s16x4 hi4 = lds_read_tr(lds + vaddr[db & 1] + imm + SUB_B * 2);
        s16x8 a;
        a[0] = lo4[0]; a[1] = lo4[1]; a[2] = lo4[2]; a[3] = lo4[3];
        a[4] = hi4[0]; a[5] = hi4[1]; a[6] = hi4[2]; a[7] = hi4[3];
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
    // ring addresses of the coming matrix phase: formed in the vector phase in front of it (for its
    // `prefill`) and kept, so that the matrix phase opens with its first MFMA
    int ck_addr = 0, cv_addr[2] = {0, 0};
    auto prefill = [&](int kslot, int vslot, auto first_, auto last_) __attribute__((always_inline)) {
        constexpr int FIRST = decltype(first_)::value, LAST = decltype(last_)::value;
        slot_addr(kslot, vslot, ck_addr, cv_addr);
#pragma unroll
        for (int i = FIRST; i < FIRST + FW; ++i)
            if (i < LAST) win[(i - FIRST) % FW] = frag(ck_addr, cv_addr, i);
    };
    auto mma_stream = [&](auto first_, auto last_, auto&& after) __attribute__((always_inline)) {
        constexpr int FIRST = decltype(first_)::value, LAST = decltype(last_)::value;
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
                if constexpr (LM) {
                    if (db == DB / 2) {
                        lacc[0] = M::run(ones, pb[st][0], lacc[0]);
                        lacc[1] = M::run(ones, pb[st][1], lacc[1]);
                    }
                }
            }
            if (i + FW < LAST) win[(i - FIRST) % FW] = frag(ck_addr, cv_addr, i + FW);
            after(i);
            // pin the order: left alone, the scheduler sinks every read to just before its MFMA
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    // ---- the item the vector phases work on
    Seq cq = {0, 0};
    Item cur = decode(cq);
    int qrow0 = cur.q0 + wave_u * 32 + i16;      // this lane's query of block 0; block 1: + 16

    // MASK = false: a tile that lies before every row of the item and inside N (causal builds run
    // those tiles in a loop of their own, whose code then is the code of the unmasked kernel).
    // An item's first tile needs nothing special: O, m and l were cleared behind the previous item,
    // and against a running max of -inf every weight and with it every row sum is infinite (or NaN
    // where the logit is masked), which sends the speculative path back and trips the other
    // path's vote; O and l, still zero, are scaled by exp2(-inf) = 0.
    auto softmax = [&](int kv0, auto mask_flag, auto first_flag) __attribute__((always_inline)) {
        constexpr bool MASK = decltype(mask_flag)::value != 0;
        [[maybe_unused]] constexpr bool FIRST = decltype(first_flag)::value != 0;   // an item's first tile
        // keys past N, and under the causal mask keys after the lane's query rows; a tile the
        // wave's first row already covers entirely needs no look
        if (MASK && (kv0 + kBKVS > N || (CAUSAL && kv0 + kBKVS - 1 > cur.q0 + 32 * wave_u))) {
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
                    for (int qb = 0; qb < 2; ++qb) {
                        s[kb][qb][r] = hide_past<16 * kb + r>(s[kb][qb][r], t_[qb], ninf);
                    }
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
                if constexpr (LM) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) lacc[qb][r] *= alpha;
                }
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
        auto weights = [&]() __attribute__((always_inline)) {
#pragma unroll
            for (int qb = 0; qb < 2; ++qb) {
#pragma unroll
                for (int kb = 0; kb < 4; ++kb)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        p[kb][qb][r] = __builtin_amdgcn_exp2f(__builtin_fmaf(s[kb][qb][r], sl2, -m[qb]));
                        if constexpr (!LM) {
                            if (kb == 0 && r == 0) psum[qb] = p[kb][qb][r];
                            else psum[qb] += p[kb][qb][r];
                        }
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
            // an item's first tile always looks for its maxima (m is -inf), and so do the masked tiles on
            // the diagonal of a causal launch (nothing is speculated on there: see SPECULATE)
            bool back = true;
            if constexpr (!FIRST && SPECULATE) {
                weights_pre(none, IC<0>{});
                // negated comparisons: a NaN or an infinite sum goes back too
                back = __builtin_amdgcn_ballot_w64(!(psum[0] <= lim) || !(psum[1] <= lim)) != 0;   // wave-uniform
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
            if constexpr (LM) {
                // The lane's largest weight per query stands in for its row sum (which the matrix cores form).
                // Plain fmaxf, not the asm forms used on MFMA results: an instruction the compiler does not
                // know as a vector instruction gets no wait state behind the v_exp_f32 whose result it
                // reads (transcendental results arrive a cycle late) and would see the register's old value.
#pragma unroll
                for (int qb = 0; qb < 2; ++qb) {
                    float a = __builtin_fmaxf(__builtin_fmaxf(p[0][qb][0], p[0][qb][1]), p[0][qb][2]);
                    float b = __builtin_fmaxf(__builtin_fmaxf(p[2][qb][0], p[2][qb][1]), p[2][qb][2]);
                    a = __builtin_fmaxf(__builtin_fmaxf(a, p[0][qb][3]), p[1][qb][0]);
                    b = __builtin_fmaxf(__builtin_fmaxf(b, p[2][qb][3]), p[3][qb][0]);
                    a = __builtin_fmaxf(__builtin_fmaxf(a, p[1][qb][1]), p[1][qb][2]);
                    b = __builtin_fmaxf(__builtin_fmaxf(b, p[3][qb][1]), p[3][qb][2]);
                    psum[qb] = __builtin_fmaxf(__builtin_fmaxf(a, b), __builtin_fmaxf(p[1][qb][3], p[3][qb][3]));
                }
            }
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
            // (the exchange between the four lanes of a row only in tiles that move the max at
            // D = 64, in front of the vote at D = 128: see attn_fwd_mfma16_kernel)
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
        if constexpr (!LM) {
            l[0] += psum[0];
            l[1] += psum[1];
        }
#pragma unroll
        for (int st = 0; st < 2; ++st)
#pragma unroll
            for (int qb = 0; qb < 2; ++qb)
#pragma unroll
                for (int j = 0; j < 8; ++j) pb[st][qb][j] = (T)p[2 * st + (j >> 2)][qb][j & 3];
        // The conversions belong to the vector phase: left alone, the compiler sinks them (and the two
        // additions to l) below the phase barrier, in front of the matrix phase's first MFMA.
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

    // The finished item's O and row statistics leave the wave: normalised, parked in the wave's
    // own LDS region 16 rows at a time and stored as whole rows (16 bytes per lane, each row
    // contiguous); then O, m, l are cleared for the next item.
    auto store_item = [&]() __attribute__((always_inline)) {
        Pair ltot;
        if constexpr (LM) ltot = Pair{lacc[0][0], lacc[1][0]};        // complete: the MFMA sums over all keys
        else ltot = four_row_reduce(l[0], l[1], SumOp{});
        const float inv[2] = {1.0f / ltot.a, 1.0f / ltot.b};
        char* obuf = lds + RING_B + wave_u * PARK_B;
        const size_t hoff = (size_t)cur.head * head_elems;
        constexpr int RPI = 64 / CPR;            // rows per store instruction
        const int orow0 = cur.q0 + wave_u * 32;
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
#pragma unroll
            for (int db = 0; db < DB; ++db) {
                vec4 w;
#pragma unroll
                for (int r = 0; r < 4; ++r) w[r] = (T)(o[db][qb][r] * inv[qb]);
                *reinterpret_cast<vec4*>(obuf + i16 * OROW_B + (16 * db + 4 * g) * 2) = w;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // own writes only: the region is per wave
#pragma unroll
            for (int k2 = 0; k2 < 16 / RPI; ++k2) {
                const int row = k2 * RPI + lane / CPR, ch = lane % CPR;
                const u32x4 val = *reinterpret_cast<const u32x4*>(obuf + row * OROW_B + ch * 16);
                const int grow = orow0 + 16 * qb + row;
                if (grow < N && ch * 8 < gd)
                    *reinterpret_cast<u32x4*>(Og + hoff + (size_t)grow * gd + ch * 8) = val;
            }
            // (the next 16 rows are parked over these: LDS serves one wave's requests in order, and
            // the fence keeps the compiler from moving the writes above the reads)
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        }
        // lanes 0-15 hold block 0's statistics for their query, lanes 16-31 write block 1's
        if (LSEg != nullptr && g < 2) {
            const int qr = qrow0 + 16 * g;
            const float mm = g == 0 ? m[0] : m[1], ll = g == 0 ? ltot.a : ltot.b;
            if (qr < N) LSEg[(size_t)cur.head * N + qr] = (mm + __builtin_amdgcn_logf(ll)) * 0.6931471805599453f;
        }
        clear_state();
    };

    // vmcnt(0), as the builtin rather than asm text: the compiler's own wait placement sees it, so it
    // knows behind it that no LDS-DMA fetch and no Q load is in flight and adds no wait of its own in
    // front of the matrix phase's reads.  (simm16: vmcnt[3:0] = 0, expcnt = 7, lgkmcnt = 15, vmcnt[5:4] = 0)
    auto dma_wait = [&]() __attribute__((always_inline)) {
        asm volatile("" ::: "memory");
        __builtin_amdgcn_s_waitcnt(0x0F70);
        asm volatile("" ::: "memory");
    };
    // Behind dma_wait() no Q load is in flight either.  The compiler does not conclude that across the
    // loops (it opened every matrix phase with five waits on the Q registers): passing the fragments
    // through an empty asm statement makes them values defined here, with no load attached.
    auto q_settled = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int qb = 0; qb < 2; ++qb)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                u32x4 w = __builtin_bit_cast(u32x4, qf[qb][ks]);
                asm volatile("" : "+v"(w));
                qf[qb][ks] = __builtin_bit_cast(vec8, w);
            }
    };
    auto phase_barrier = [&]() __attribute__((always_inline)) {
        asm volatile("" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };
    // Behind the matrix phase that formed the LAST S of item x (tile T-1): the Q fragments are free,
    // the next item's go into them (they have the following vector phase to arrive).
    auto q_turnover = [&](Seq nq) __attribute__((always_inline)) {
        if (nq.ps < my_units) {
            load_q(decode(nq));
            q_fresh = 1;
        }
    };
    // behind dma_wait(): fragments that arrived in this phase get their scale (PRE)
    auto q_arrived = [&]() __attribute__((always_inline)) {
        if constexpr (PRE) {
            if (q_fresh) {
                scale_q();
                q_fresh = 0;
            }
        }
        q_settled();
    };

    // ---- prologue: the first item's Q fragments into their registers, this workgroup's first three
    // tiles into their ring slots.  S(0) needs Q and K(0) only; V(0) and K(1) are asked for behind the
    // barrier and arrive under S(0) and the first vector phase.  (The wait is for ALL vector-memory
    // operations, not a count: register spills, where a build has them, count in vmcnt as well.)
    load_q(cur);
#pragma unroll
    for (int i = 0; i < LPT; ++i) fetch_piece(kc, 0, i);
    advance(kc, Kg);
    dma_wait();
    if constexpr (PRE) scale_q();
    __syncthreads();
#pragma unroll
    for (int i = 0; i < LPT; ++i) fetch_piece(vc, 4 * TILE_B, i);
    advance(vc, Vg);
#pragma unroll
    for (int i = 0; i < LPT; ++i) fetch_piece(kc, TILE_B, i);
    advance(kc, Kg);

    prefill(0, 0, IC<0>{}, IC<NQ>{});
    // S(0) = K(0) Q^T, with the fetches of the stream's tiles K(2), V(1) behind its last fragment request
    mma_stream(IC<0>{}, IC<NQ>{}, [&](int i) __attribute__((always_inline)) {
        constexpr int L0 = NQ - FW;
#pragma unroll
        for (int j = 0; j < LPT; ++j) {
            if (i == L0 + 2 * j)     fetch_piece(kc, 2 * TILE_B, j);
            if (i == L0 + 2 * j + 1) fetch_piece(vc, (4 + 1) * TILE_B, j);
        }
    });
    advance(kc, Kg);
    advance(vc, Vg);
    if (cur.nt == 1) q_turnover(seq_next(cq, cur.pair2));
    if (grp) __syncthreads();
    softmax(0, IC<1>{}, IC<1>{});
    dma_wait();                              // every wave's share of the five tiles is in before the barrier
    q_arrived();
    prefill(1, 0, IC<0>{}, IC<NQ + NP>{});   // first fragments of the next matrix phase: K of stream tile 1
    phase_barrier();

    // Ring bookkeeping (wave-uniform): stream tile u of K lives in slot u % 4, of V in slot u % 3.
    // The fetches of K(u+3) and V(u+2) are issued under the last MFMAs of M(u), behind the phase's
    // last fragment request, and every wave waits for its own at the end of its next vector phase.
    int k1 = 1, v0 = 0;              // (u+1) % 4, u % 3
    // One matrix phase: S of the stream's next tile, O += V^T P^T of the current one, two fetches.
    // The matrix phase is the critical path of the pair of phases between two barriers, so nothing
    // follows its last MFMA but the barrier: the cursors move on (`turn`) at the head of the vector
    // phase behind it.
    auto matrix_phase = [&]() __attribute__((always_inline)) {
        const int k3 = (k1 + 2) & 3;             // (u+3) % 4
        const int v1 = v0 == 2 ? 0 : v0 + 1;     // (u+1) % 3
        const int v2 = v1 == 2 ? 0 : v1 + 1;     // (u+2) % 3
        mma_stream(IC<0>{}, IC<NQ + NP>{}, [&](int i) __attribute__((always_inline)) {
            constexpr int L0 = NQ + NP - FW;     // no fragment requests from here on
#pragma unroll
            for (int j = 0; j < LPT; ++j) {
                if (i == L0 + 2 * j)     fetch_piece(kc, k3 * TILE_B, j);
                if (i == L0 + 2 * j + 1) fetch_piece(vc, (4 + v2) * TILE_B, j);
            }
        });
    };
    auto turn = [&]() __attribute__((always_inline)) {
        advance(kc, Kg);
        advance(vc, Vg);
    };
    // The tail of a vector phase: this wave's share of the tiles fetched a phase ago is in LDS; the
    // first fragments of the next matrix phase are requested; the ring indices move on.
    auto phase_tail = [&]() __attribute__((always_inline)) {
        const int k2 = (k1 + 1) & 3;
        const int v1 = v0 == 2 ? 0 : v0 + 1;
        dma_wait();
        q_arrived();
        prefill(k2, v1, IC<0>{}, IC<NQ + NP>{});
        phase_barrier();
        k1 = k2; v0 = v1;
    };
    auto tile_step = [&](int t, auto mask_flag) __attribute__((always_inline)) {
        matrix_phase();                          // M(t): S(t+1), O += V(t)^T P(t)^T
        phase_barrier();                         // the fetches just issued stay in flight across it
        turn();
        if (t + 2 == cur.nt) q_turnover(seq_next(cq, cur.pair2));   // S(T-1) is formed: the Q registers turn over
        softmax((t + 1) * kBKVS, mask_flag, IC<0>{});   // X(t+1)
        phase_tail();
    };

#pragma nounroll
    for (;;) {
        int t = 0;
        if constexpr (CAUSAL) {
            // tiles 1 .. t_clean-1 lie before the item's first row and inside N: no mask code
            const int t_clean = min(cur.q0, N) / kBKVS;
#pragma nounroll
            for (; t + 1 < min(cur.nt, t_clean); ++t) tile_step(t, IC<0>{});
        }
#pragma nounroll
        for (; t + 1 < cur.nt; ++t) tile_step(t, IC<1>{});
        // ---- the item's last tile: M(T-1) forms S(0) of the next item beside O += V(T-1)^T P^T
        // (behind the last item the S half works on whatever the ring holds; nobody reads it)
        const Seq nq = seq_next(cq, cur.pair2);
        const bool more = nq.ps < my_units;
        const Item nxt = more ? decode(nq) : cur;
        matrix_phase();
        if (!more) dma_wait();                   // no fetch may still be writing LDS when the wave ends
        phase_barrier();
        turn();
        if (more && nxt.nt == 1) q_turnover(seq_next(nq, nxt.pair2));
        store_item();
        if (!more) break;
        cur = nxt;
        cq = nq;
        qrow0 = cur.q0 + wave_u * 32 + i16;
        softmax(0, IC<1>{}, IC<1>{});            // X(0) of the new item
        phase_tail();
    }
    if (grp == 0) __syncthreads();   // the leading half meets the lagging half's last barrier
}

int g_cu_count[64] = {};

// Compute units of the current device (cached per device; 0 on failure, with the error recorded).
int cu_count_current() {
    int dev = -1;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess || dev < 0 || dev >= 64) {
        fail(MYFA_E_DEVICE, "attention: no usable current device (%s)", hipGetErrorString(e));
        return 0;
    }
    int n = __atomic_load_n(&g_cu_count[dev], __ATOMIC_RELAXED);
    if (n == 0) {
        e = hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
        if (e != hipSuccess || n <= 0) {
            fail(MYFA_E_DEVICE, "attention: cannot read the device's compute-unit count (%s)", hipGetErrorString(e));
            return 0;
        }
        __atomic_store_n(&g_cu_count[dev], n, __ATOMIC_RELAXED);
    }
    return n;
}

template <typename T, int D>
int launch_stream(const void* q, const void* k, const void* v, void* o, float* lse,
                  int64_t heads, int64_t N, int64_t d, float scale, float thr, bool spec, bool pre, bool lm, bool causal, int wgs, bool wide, hipStream_t s) {
    if (N * d * 2 >= (int64_t)kPastEndS) return fail(MYFA_E_UNSUPPORTED, "attention: N*d=%lld per head too large", (long long)(N * d));
    const int64_t nqblk = cdiv(N, kBQS);
    // units of work: query blocks, or under the causal mask pairs of mirrored query blocks of a head
    const int64_t hp = causal ? (nqblk + 1) / 2 : nqblk;
    const int64_t nunits = heads * hp;
    if (heads * nqblk > 0x7fffffffLL) return fail(MYFA_E_UNSUPPORTED, "attention: too many query blocks (%lld)", (long long)(heads * nqblk));
    // as many workgroups as the chip holds at once (one per CU at D = 128, two at D = 64), fewer if
    // there are fewer units
    int64_t cap = wgs;
    if (cap <= 0) {
        const int cus = cu_count_current();
        if (cus == 0) return MYFA_E_DEVICE;
        cap = (int64_t)cus * (D == 64 && !wide ? 2 : 1);
    }
    const int64_t grid = nunits < cap ? nunits : cap;
    const float sl2 = scale * 1.4426950408889634f;
    const float lim = exp2f(thr);   // the speculative path compares row sums, not logits
    auto go = [&](auto kern) {
        launch_kernel(kern, dim3((unsigned)grid), dim3(kThreadsS), 0, s,
                      static_cast<const T*>(q), static_cast<const T*>(k), static_cast<const T*>(v),
                      static_cast<T*>(o), lse, (int)N, (int)nqblk, (int)hp, (int)nunits, sl2, thr, lim, (int)d);
    };
    if constexpr (D == 64) {
        if (wide) {
            if (causal && spec && pre) go(attn_fwd_stream_kernel<T, D, true, true, true, true>);
            else if (causal) go(attn_fwd_stream_kernel<T, D, true, false, false, true>);
            else if (spec && pre) go(attn_fwd_stream_kernel<T, D, false, true, true, true>);
            else if (spec) go(attn_fwd_stream_kernel<T, D, false, true, false, true>);
            else go(attn_fwd_stream_kernel<T, D, false, false, false, true>);
            return check_launch("attn_fwd_stream");
        }
    }
    if constexpr (D == 128) {
        if (!causal && spec && !pre && lm) {
            go(attn_fwd_stream_kernel<T, D, false, true, false, false, true>);
            return check_launch("attn_fwd_stream");
        }
    }
    if (causal && spec && pre) go(attn_fwd_stream_kernel<T, D, true, true, true>);
    else if (causal) go(attn_fwd_stream_kernel<T, D, true, false>);
    else if (spec && pre) go(attn_fwd_stream_kernel<T, D, false, true, true>);
    else if (spec) go(attn_fwd_stream_kernel<T, D, false, true>);
    else go(attn_fwd_stream_kernel<T, D, false, false>);
    return check_launch("attn_fwd_stream");
}

}  // namespace

// dtype: MYFA_F16 or MYFA_BF16; d: a multiple of 8 in 8..128; heads = B*H.
// spec, thr: as attn_fwd_mfma16 (attn_fwd_mfma16.hip): < 0 = the library's choice.
// wgs: workgroups in the grid; <= 0 = as many as the chip holds at once.  occ: see below.
int attn_fwd_stream(const void* q, const void* k, const void* v, void* o, float* lse,
                    int64_t heads, int64_t N, int64_t d, float scale, float thr, int spec, int pre, int wgs, int occ,
                    int dtype, bool causal, hipStream_t s) {
    if (heads <= 0 || N <= 0) return MYFA_OK;
    // occ (head dims up to 64): 1 = one workgroup per CU with 256 registers per lane, 2 = two with 128,
    // else the library's choice: one.  (Two spill 40-110 vector registers; one, under the causal mask,
    // beats the one-block-per-workgroup kernel by 8-14 % at B4 H32 N4096 d64 and 10 % at B8 H16 N1024
    // d64, and without the mask loses 6-9 % to it -- the dispatch sends only causal launches here.)
    const bool wide = d <= 64 && occ != 2;
    // spec: 0 = the max of every tile and a vote; 1 = speculative; 2 = speculative with the row sums
    // formed by the matrix cores (head dims above 64, no mask); < 0 = the library's choice: 1 where
    // the logits of unit-variance inputs would spread by at most 4 base-2 units; beyond that 2 for
    // bfloat16 (whose weights have the exponent range to sit 2^60 above a stale max: on such logits
    // the max of a row stops moving by that much after its first tiles; measured +12 % over the
    // vote at scale = 1, d = 128) and 0 for float16, whose weights overflow at 2^16.
    const bool small_logits = scale * 1.4426950408889634f * sqrtf((float)d) <= 4.0f;
    // (Not by itself when the row statistic is asked for: a sum of weights rounded to 8 bits moves
    // log-sum-exp by up to 2^-9, and a backward recomputes P against exactly that statistic.)
    // (2 where that build does not exist -- head dims up to 64, the causal mask -- is the library's choice.)
    const bool lm_ok = d > 64 && !causal;
    const bool choose = spec < 0 || (spec == 2 && !lm_ok);
    const bool lm = lm_ok && (spec == 2 || (choose && !small_logits && dtype == MYFA_BF16 && lse == nullptr));
    const bool sp = choose ? small_logits || lm : spec != 0;
    // pre: < 0 = the library's choice: wherever the speculative build runs and no row statistic is
    // asked for (inference).  The rounding of Q * scale * log2(e) moves a logit by ~2^-9 of its size --
    // the order of the rounding the 16-bit inputs already carry, and far inside the output's bound --
    // but the row log-sum-exp moves with its largest logits (0.01 for a logit of 35), and a backward
    // that recomputes the logits from the unscaled Q wants the statistic of exactly those logits.
    // Else forced on / off (tuning flavour).
    // (Head dims up to 64 with two workgroups per CU: never by itself -- that build already spills
    // registers and loses a third with it.)
    const bool pr = sp && !lm && (pre < 0 ? lse == nullptr && (d > 64 || wide) : pre != 0);
    if (thr < 0.f) thr = !sp || (causal && !pr) ? 8.f : dtype != MYFA_BF16 ? 12.f : lm ? 60.f : 30.f;
    if (dtype == MYFA_F16 && thr > 15.f) thr = 15.f;
    if (thr > 60.f) thr = 60.f;
    // head dims that are multiples of 8 run on the next build up, the missing columns read as zeros
    if (d >= 8 && d <= 128 && d % 8 == 0) {
        const bool big = d > 64;
        if (dtype == MYFA_BF16) return big ? launch_stream<__bf16, 128>(q, k, v, o, lse, heads, N, d, scale, thr, sp, pr, lm, causal, wgs, wide, s)
                                           : launch_stream<__bf16, 64>(q, k, v, o, lse, heads, N, d, scale, thr, sp, pr, lm, causal, wgs, wide, s);
        if (dtype == MYFA_F16)  return big ? launch_stream<_Float16, 128>(q, k, v, o, lse, heads, N, d, scale, thr, sp, pr, lm, causal, wgs, wide, s)
                                           : launch_stream<_Float16, 64>(q, k, v, o, lse, heads, N, d, scale, thr, sp, pr, lm, causal, wgs, wide, s);
    }
    return fail(MYFA_E_UNSUPPORTED, "attention (f16/bf16): head dim %lld is not a multiple of 8 in 8..128", (long long)d);
}

}  // namespace myfa
