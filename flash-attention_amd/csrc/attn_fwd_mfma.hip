// Fused attention forward for f16 / bf16 on gfx950 matrix cores.
//
//     O[b,h] = softmax(scale * Q[b,h] K[b,h]^T) V[b,h]      Q,K,V,O: [B,H,N,D], D in {64,128}
// (head dims above 128: attn_fwd_mfma256.hip)
//
// This is the kernel the reference sketches and leaves as a TODO
// (csrc/my_flash_attn/flash_attention.cu:23-69).  The math is the online-softmax
// recurrence of its Python restatement (tests/test_flash_attn.py:82-93: running
// rowmax m, rowsum l, rescale by exp(m_old - m_new)), in the loop order that keeps
// a Q tile resident and streams K/V tiles past it, with O left un-normalised in
// registers until one final division by l.
//
// Two kernels in this file, both on v_mfma_f32_32x32x16 (the 16x16x32 build of the same design is
// attn_fwd_mfma16.hip; attn_fwd_dispatch.hip picks between them):
//   attn_fwd_mfma_pp_kernel  the first ping-pong build (only the tuning flavour can select it): K/V tiles fetched by LDS-DMA
//                            (buffer_load ... lds) into rings of 4 / 3 slots, the two waves of a
//                            SIMD in opposite phases (matrix / softmax), one s_barrier per phase.
//                            Described in the comment in front of it.
//   attn_fwd_mfma_kernel     the first schedule, kept for cross-checks (only the tuning flavour can select it): tiles
//                            staged through registers (global_load_dwordx4 one tile ahead,
//                            ds_write_b128 behind the MFMAs) into a double buffer with per-row XOR
//                            swizzles (k_off / v_off below), one barrier per tile.
//
// Work decomposition (both)
//   workgroup = 8 waves = 256 query rows of one (batch, head); wave w owns rows 32w..32w+31.
//   K/V are streamed in tiles of 64 keys.
//
// Register choreography (v_mfma_f32_32x32x16: D[32x32] += A[32x16] B[16x32])
//   S^T = K Q^T : A = K rows (keys), B = Q^T, so the accumulator has the QUERY on
//   the lane (col = lane & 31) and 16 keys in its registers; the two lane halves
//   hold different keys of the same query.  Row max / row sum are therefore
//   in-lane loops plus one half<->half exchange (v_permlane32_swap).
//   O^T = V^T P^T : the S^T accumulator, converted to 16-bit in place, is already
//   the B operand of this product (it sums over the accumulator's ROW index, the
//   key), with the k order inside a 16-key step permuted to
//   key = 16s + 8(j>>2) + 4(lane>>5) + (j&3) for element j; the A operand V^T is
//   read from the row-major V tile with the transposing LDS read
//   ds_read_b64_tr_b16 in that same key order.  O^T again has the query on the
//   lane, so the per-row rescale and the final 1/l are per-lane scalars.
// Compiled into the tuning flavour of the library only (-DMYFA_TUNING): the shipped library's
// dispatch (attn_fwd_dispatch.hip) never selects these kernels.
#ifdef MYFA_TUNING
#include "attn_mfma_common.h"
#include "tuning.h"

namespace myfa {
namespace {

template <int D> __device__ __forceinline__ int k_off(int row, int ch) {
    if constexpr (D == 128) return row * 256 + ((ch ^ (row & 15)) << 4);
    else                    return row * 128 + ((ch ^ ((row >> 1) & 7)) << 4);
}
template <int D> __device__ __forceinline__ int v_off(int row, int ch) {
    if constexpr (D == 128) return row * 256 + ((ch ^ (((row & 3) << 2) | ((row >> 2) & 3))) << 4);
    else                    return row * 128 + ((ch ^ ((((row >> 1) & 1) << 2) | ((row >> 2) & 3))) << 4);
}

constexpr int kWaves = 8;
constexpr int kThreads = kWaves * 64;
constexpr int kBQ = kWaves * 32;   // query rows per workgroup
constexpr int kBKV = 64;           // keys per tile
// A buffer offset no tensor reaches (the launcher keeps a head below it) and that a tile offset
// added to it cannot wrap: fetches from there read zero without touching memory.
constexpr int kPastEndV1 = 0x7FFF0000;

template <typename T, int D>
__global__ void __launch_bounds__(kThreads, 2)
attn_fwd_mfma_kernel(const T* __restrict__ Qg, const T* __restrict__ Kg, const T* __restrict__ Vg,
                     T* __restrict__ Og, float* __restrict__ LSEg, int N, int nqblk, int nwg, float sl2) {
    using M = Mma<T>;
    using vec8 = typename M::vec8;
    using vec4 = typename M::vec4;
    constexpr int CPR = D / 8;                   // 16-byte chunks per row
    constexpr int TILE_B = kBKV * D * 2;         // bytes per K (or V) tile
    constexpr int KS = D / 16;                   // k-steps of S^T = K Q^T
    constexpr int DB = D / 32;                   // 32-row blocks of O^T
    constexpr int LPT = kBKV * CPR / kThreads;   // 16-byte loads per thread per tile
    static_assert(LPT >= 1, "tile smaller than one load per thread");

    __shared__ __attribute__((aligned(16))) char lds[4 * TILE_B];   // K0 K1 V0 V1

    // Workgroups that share one head's K/V get consecutive logical ids, and the
    // remap hands each XCD (blockIdx % 8 under round-robin dispatch) a contiguous
    // range of logical ids, so a head's K/V stay in one XCD's L2.  Speed only.
    int lid;
    {
        const int bid = blockIdx.x, xcd = bid & 7, idx = bid >> 3, qq = nwg >> 3, rr = nwg & 7;
        lid = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + idx;
    }
    const int head = lid / nqblk;
    const int q0 = (lid - head * nqblk) * kBQ;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;

    const size_t hoff = (size_t)head * (size_t)N * D;
    const T* Qh = Qg + hoff;
    const T* Kh = Kg + hoff;
    const T* Vh = Vg + hoff;

    // ---- Q^T fragments: lane (r, h) holds Q[row][16ks + 8h .. +7] for ks = 0..KS-1
    const int qrow = q0 + wave * 32 + r;
    const int qrow_c = qrow < N ? qrow : N - 1;
    vec8 qf[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
        qf[ks] = *reinterpret_cast<const vec8*>(Qh + (size_t)qrow_c * D + 16 * ks + 8 * h);

    // ---- staging of one K/V tile through registers
    u32x4 kst[LPT], vst[LPT];
    auto stage_load = [&](int kv0) {
#pragma unroll
        for (int i = 0; i < LPT; ++i) {
            const int id = tid + kThreads * i;
            const int row = id / CPR, ch = id % CPR;
            int g = kv0 + row;
            g = g < N ? g : N - 1;                // tail rows: any valid row; masked later
            kst[i] = *reinterpret_cast<const u32x4*>(Kh + (size_t)g * D + ch * 8);
            vst[i] = *reinterpret_cast<const u32x4*>(Vh + (size_t)g * D + ch * 8);
        }
    };
    auto stage_write = [&](int buf) {
#pragma unroll
        for (int i = 0; i < LPT; ++i) {
            const int id = tid + kThreads * i;
            const int row = id / CPR, ch = id % CPR;
            *reinterpret_cast<u32x4*>(lds + buf * TILE_B + k_off<D>(row, ch)) = kst[i];
            *reinterpret_cast<u32x4*>(lds + (2 + buf) * TILE_B + v_off<D>(row, ch)) = vst[i];
        }
    };

    f32x16 o[DB];
#pragma unroll
    for (int db = 0; db < DB; ++db)
#pragma unroll
        for (int i = 0; i < 16; ++i) o[db][i] = 0.f;
    float m = -INFINITY;   // running row max of scale*log2(e)*S
    float l = 0.f;         // this lane half's share of the running row sum

    const int ntiles = (N + kBKV - 1) / kBKV;
    stage_load(0);
    stage_write(0);
    __syncthreads();

    // lane-constant parts of the transposing-read address
    const int tq = (lane & 15) >> 2, tp = lane & 3, tg1 = (lane >> 4) & 1;

    for (int t = 0; t < ntiles; ++t) {
        const int buf = t & 1;
        const int kv0 = t * kBKV;
        // Always issued (the row index is clamped inside): a load under a condition makes
        // hipcc guard the staging registers with conservative s_waitcnt vmcnt(0).
        stage_load(kv0 + kBKV);

        const char* Kb = lds + buf * TILE_B;
        const char* Vb = lds + (2 + buf) * TILE_B;

        // ---- S^T = K Q^T : 2 key blocks x KS k-steps
        f32x16 s[2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
            for (int i = 0; i < 16; ++i) s[kb][i] = 0.f;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const s16x8 a = *reinterpret_cast<const s16x8*>(Kb + k_off<D>(32 * kb + r, 2 * ks + h));
                s[kb] = M::run(a, qf[ks], s[kb]);
            }
        }

        // ---- mask keys past N (last tile only)
        if (kv0 + kBKV > N) {
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int key = kv0 + 32 * kb + (i & 3) + 8 * (i >> 2) + 4 * h;
                    if (key >= N) s[kb][i] = -INFINITY;
                }
        }

        // ---- online softmax (per query = per lane, two halves cooperate)
        float tmax = s[0][0];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int i = 0; i < 16; ++i) tmax = fmaxf(tmax, s[kb][i]);
        tmax = half_exchange_max(tmax);
        const float m_new = fmaxf(m, tmax * sl2);
        const float alpha = __builtin_amdgcn_exp2f(m - m_new);
        m = m_new;
        float psum = 0.f;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                s[kb][i] = __builtin_amdgcn_exp2f(__builtin_fmaf(s[kb][i], sl2, -m_new));
                psum += s[kb][i];
            }
        l = l * alpha + psum;
#pragma unroll
        for (int db = 0; db < DB; ++db)
#pragma unroll
            for (int i = 0; i < 16; ++i) o[db][i] *= alpha;

        // ---- O^T += V^T P^T : 4 k-steps of 16 keys x DB row blocks
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int st = 0; st < 2; ++st) {
                vec8 pb;
#pragma unroll
                for (int j = 0; j < 8; ++j) pb[j] = (T)s[kb][8 * st + j];
                const int r0 = 32 * kb + 16 * st + 4 * h + tq;   // key row of the first 4-row block
#pragma unroll
                for (int db = 0; db < DB; ++db) {
                    const int ch = 4 * db + 2 * tg1 + (tp >> 1);
                    const s16x4 lo = lds_read_tr(Vb + v_off<D>(r0, ch) + 8 * (tp & 1));
                    const s16x4 hi = lds_read_tr(Vb + v_off<D>(r0 + 8, ch) + 8 * (tp & 1));
                    s16x8 a;
                    a[0] = lo[0]; a[1] = lo[1]; a[2] = lo[2]; a[3] = lo[3];
                    a[4] = hi[0]; a[5] = hi[1]; a[6] = hi[2]; a[7] = hi[3];
                    o[db] = M::run(a, pb, o[db]);
                }
            }

        // ---- publish the next tile: its buffer was last read before the previous barrier
        stage_write(buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue: O = O^T / l, written as rows; lane (r,h) holds, for its query,
    // d = 32db + 8g + 4h + (0..3) in registers 4g..4g+3 of block db
    const float ltot = half_exchange_sum(l);
    const float inv = 1.0f / ltot;
    if (qrow < N) {
        T* orow = Og + hoff + (size_t)qrow * D;
#pragma unroll
        for (int db = 0; db < DB; ++db)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                vec4 w;
#pragma unroll
                for (int j = 0; j < 4; ++j) w[j] = (T)(o[db][4 * g + j] * inv);
                *reinterpret_cast<vec4*>(orow + 32 * db + 8 * g + 4 * h) = w;
            }
        if (LSEg != nullptr && h == 0)
            LSEg[(size_t)head * N + qrow] = (m + __builtin_amdgcn_logf(ltot)) * 0.6931471805599453f;
    }
}


// ---------------------------------------------------------------------------
// Ping-pong variant (the default).  Same tiles and register choreography as the
// kernel above; different schedule and a different LDS image.
//
// Schedule.  The loop is software-pipelined into a matrix phase
//   M(t) = { S(t+1) = K(t+1) Q^T ;  O += V(t)^T P(t)^T }      (32 MFMAs back to back)
// and a vector phase
//   X(t+1) = { online softmax of S(t+1) -> P(t+1), rescale O },
// and the two waves that share a SIMD (wave w and wave w+4) run in opposite
// phases, one barrier per phase: while waves 0-3 keep the matrix pipe busy,
// waves 4-7 do their exponentials on the vector pipe, and vice versa.  Waves 4-7
// simply run the same program one phase late.
//

//
// Instruction economy.  Both waves of a SIMD issue their vector AND matrix
// instructions through the same port, so every address computation in the loop
// is paid for twice.  Hence:
//  * one LDS image for K and V tiles, made of 8-row x 64-byte subtiles,
//        off(row, c) = 16*D*(row>>3) + 512*(c>>2) + 64*(row&7) + 16*((c&3) ^ ((row>>2)&3))
//    (c = 16-byte chunk of the row).  Both the row reads of K (ds_read_b128) and the
//    transposing reads of V (ds_read_b64_tr_b16) are bank-conflict-free on it, and
//    every fragment address is one of four per-lane bases plus a compile-time
//    immediate, so the matrix phase computes no addresses at all (the loop is
//    unrolled by two to make the slot a compile-time constant);
//  * K/V tiles are fetched with LDS-DMA buffer loads: per-lane source offset fixed, tile
//    offset in a scalar register, rows past N return zero (no clamping arithmetic, no
//    staging registers, no ds_write pass).
//
// `thr` is the deferred-max threshold in log2 units: the running max only moves
// (and O, l are only rescaled) in a tile where some row's max grew by more than
// thr.  thr = 0 rescales whenever any max grew, which is exactly the textbook
// recurrence; thr > 0 lets P reach 2^thr.

// STAMP builds are diagnostic only: they bracket the phases with s_memtime and write the
// per-phase cycle sums of waves 0 and 4 to `dbg` (24 u64 per workgroup); never timed.
__device__ __forceinline__ unsigned long long stamp() {
    __builtin_amdgcn_sched_barrier(0);
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
    __builtin_amdgcn_sched_barrier(0);
    return t;
}



template <typename T, int D, bool STAMP = false, bool CAUSAL = false>
__global__ void __launch_bounds__(kThreads, D == 64 ? 4 : 2)
attn_fwd_mfma_pp_kernel(const T* __restrict__ Qg, const T* __restrict__ Kg, const T* __restrict__ Vg,
                        T* __restrict__ Og, float* __restrict__ LSEg, int N, int nqblk, int nwg,
                        float sl2, float thr, int gd, unsigned long long* __restrict__ dbg = nullptr) {
    using M = Mma<T>;
    using vec8 = typename M::vec8;
    using vec4 = typename M::vec4;
    constexpr int CPR = D / 8;
    constexpr int TILE_B = kBKV * D * 2;
    constexpr int SUB_B = 16 * D;                // bytes of one 8-row group of the image
    constexpr int KS = D / 16;
    constexpr int DB = D / 32;
    constexpr int LPT = kBKV * CPR / kThreads;
    static_assert(LPT >= 1, "tile smaller than one load per thread");

    __shared__ __attribute__((aligned(16))) char lds[7 * TILE_B];   // K ring of 4, V ring of 3

    int lid;
    {
        const int bid = blockIdx.x, xcd = bid & 7, idx = bid >> 3, qq = nwg >> 3, rr = nwg & 7;
        lid = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + idx;
    }
    const int head = lid / nqblk;
    // causal: a query block needs the key tiles up to its own last row only, so blocks differ
    // in length; the long ones (high rows) of each head are started first
    const int qblk = CAUSAL ? nqblk - 1 - (lid - head * nqblk) : lid - head * nqblk;
    const int q0 = qblk * kBQ;

    unsigned long long t_kernel0 = 0, t_loop0 = 0, t_loop1 = 0;
    if constexpr (STAMP) t_kernel0 = stamp();
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int grp = __builtin_amdgcn_readfirstlane(wave >> 2);   // 0: leads, 1: one phase behind
    const int r = lane & 31, h = lane >> 5;

    // gd <= D is the head dim in memory (a multiple of 8): rows are gd elements apart, and the
    // columns gd .. D-1 of the LDS images are zeros (their chunks are fetched from past the end of
    // the tensor, where the buffer bounds check answers zero).
    const size_t hoff = (size_t)head * (size_t)N * gd;

    const int qrow = q0 + wave * 32 + r;
    vec8 qf[KS];

    // ---- K/V tiles go HBM/L2 -> LDS directly (buffer_load ... lds): no staging registers, no
    // ds_write pass.  One wave-instruction fills 1 KiB of LDS, lane L at base + 16 L, so the
    // image is produced by choosing each lane's SOURCE chunk: piece `pc` of a tile covers
    // image bytes [1024 pc, 1024 pc + 1024); lane L owns 16-byte unit u = 64 pc + L, which by
    // the image formula is (row, chunk) below.  Rows past N read as zero (buffer bounds).
    const unsigned head_bytes = (unsigned)N * gd * 2;
    [[maybe_unused]] const int tile_gb = kBKV * gd * 2;            // bytes of one 64-row tile in memory
    const __amdgpu_buffer_rsrc_t krs = __builtin_amdgcn_make_buffer_rsrc((void*)(Kg + hoff), 0, head_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t vrs = __builtin_amdgcn_make_buffer_rsrc((void*)(Vg + hoff), 0, head_bytes, 0x00020000);
    [[maybe_unused]] const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    int st_goff[LPT];
#pragma unroll
    for (int i = 0; i < LPT; ++i) {
        const int u = 64 * (wave * LPT + i) + lane;
        const int grp8 = u / (SUB_B / 16), within = u % (SUB_B / 16);
        const int row = 8 * grp8 + ((within & 31) >> 2);
        const int ch = 4 * (within >> 5) + ((within & 3) ^ ((row >> 2) & 3));
        st_goff[i] = ch * 8 < gd ? (row * gd + ch * 8) * 2 : kPastEndV1;
    }
    // (The host pass of hipcc silently drops a kernel whose body names this builtin, and the
    // launch stub with it; the body only exists in the device pass.)
    // Under the causal mask the tiles behind the workgroup's last one exist but are never read:
    // they are asked for as the tile past the end, which the bounds check answers with zeros
    // without touching memory (the unconditional fetches at the end of the loop would otherwise
    // load two or three real tiles per workgroup, and the epilogue would wait for them).
    [[maybe_unused]] const int t_past = (N + kBKV - 1) / kBKV;
    [[maybe_unused]] const int t_stop = CAUSAL ? min(t_past, (q0 + kBQ) / kBKV) : t_past;
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
    // K row read, key block kb, k-step ks: row = 32kb + r, chunk = 2ks + h
    int kbase[2];
#pragma unroll
    for (int e = 0; e < 2; ++e)
        kbase[e] = SUB_B * (r >> 3) + 64 * (r & 7) + 16 * ((2 * e + h) ^ ((r >> 2) & 3));
    // V transposing read: row = 32kb + 16st + 8hi + 4h + tq, chunk = 4db + 2tg1 + (tp>>1), half tp&1
    const int tq = (lane & 15) >> 2, tp = lane & 3, tg1 = (lane >> 4) & 1;
    int vbase[2];
#pragma unroll
    for (int hi = 0; hi < 2; ++hi)
        vbase[hi] = 64 * (4 * h + tq) + 16 * ((2 * tg1 + (tp >> 1)) ^ (2 * hi + h)) + 8 * (tp & 1);

    f32x16 o[DB];
#pragma unroll
    for (int db = 0; db < DB; ++db)
#pragma unroll
        for (int i = 0; i < 16; ++i) o[db][i] = 0.f;
    float m = -INFINITY, l = 0.f;
    f32x16 s[2];
    vec8 pb[4];

    constexpr int NQ = 2 * KS, NP = 4 * DB, FW = D == 64 ? 4 : 8;
    // fragment i of the matrix phase: i < NQ: K fragment (kb = i / KS, ks = i % KS) of K slot
    // KSLOT; otherwise V fragment (kb, st, db) of V slot VSLOT.  `i` is a constant after
    // unrolling, so each address is a per-lane base plus an immediate.
    auto frag = [&](int kbyte, int vbyte, int i) __attribute__((always_inline)) -> s16x8 {
        if (i < NQ) {
            const int kb = i / KS, ks = i % KS;
            const int imm = SUB_B * 4 * kb + 512 * (ks >> 1);
            return *reinterpret_cast<const s16x8*>(lds + (kbase[ks & 1] + kbyte) + imm);
        }
        const int j = i - NQ, kb = j / (2 * DB), st = (j / DB) & 1, db = j % DB;
        const int imm = SUB_B * (4 * kb + 2 * st) + 512 * db;
        const s16x4 lo = lds_read_tr(lds + (vbase[0] + vbyte) + imm);
        const s16x4 hi = lds_read_tr(lds + (vbase[1] + vbyte) + imm + SUB_B);
        s16x8 a;
        a[0] = lo[0]; a[1] = lo[1]; a[2] = lo[2]; a[3] = lo[3];
        a[4] = hi[0]; a[5] = hi[1]; a[6] = hi[2]; a[7] = hi[3];
        return a;
    };
    // The matrix phase is one software-pipelined stream of MFMAs: the LDS fragment of MFMA
    // i + FW is requested when MFMA i issues (an LDS read issued right before its MFMA costs
    // the whole LDS latency per MFMA).  The first FW requests (`prefill`) are made at the end
    // of the PREVIOUS vector phase, before the phase barrier, so the stream starts without
    // a bubble; the window registers carry them across the barrier.
    // [FIRST, LAST) selects QK only, PV only, or both; kslot / vslot are ring slots.
    s16x8 win[FW];
    auto prefill = [&](int kslot, int vslot, auto first_, auto last_) __attribute__((always_inline)) {
        constexpr int FIRST = decltype(first_)::value, LAST = decltype(last_)::value;
        const int kbyte = kslot * TILE_B, vbyte = (4 + vslot) * TILE_B;
#pragma unroll
        for (int i = FIRST; i < FIRST + FW; ++i)
            if (i < LAST) win[(i - FIRST) % FW] = frag(kbyte, vbyte, i);
    };
    // `after(i)` runs right behind MFMA i (used to tuck the LDS-DMA issues of the next tiles
    // under the last MFMAs, where no fragment request is left to issue).
    auto mma_stream = [&](int kslot, int vslot, auto first_, auto last_, auto&& after) __attribute__((always_inline)) {
        constexpr int FIRST = decltype(first_)::value, LAST = decltype(last_)::value;
        const int kbyte = kslot * TILE_B, vbyte = (4 + vslot) * TILE_B;
        if constexpr (FIRST < NQ) {
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int i = 0; i < 16; ++i) s[kb][i] = 0.f;
        }
#pragma unroll
        for (int i = FIRST; i < LAST; ++i) {
            const s16x8 a = win[(i - FIRST) % FW];
            if (i < NQ) s[i / KS] = M::run(a, qf[i % KS], s[i / KS]);
            else o[(i - NQ) % DB] = M::run(a, pb[(i - NQ) / DB], o[(i - NQ) % DB]);
            if (i + FW < LAST) win[(i - FIRST) % FW] = frag(kbyte, vbyte, i + FW);
            after(i);
            // pin the order: left alone, the scheduler sinks every read to just before its MFMA
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    auto nothing = [](int) __attribute__((always_inline)) {};

    [[maybe_unused]] unsigned long long cyc_x1 = 0, cyc_x2 = 0;
    auto softmax = [&](int kv0) __attribute__((always_inline)) {
        unsigned long long s1 = 0, s2 = 0;
        // keys past N, and under the causal mask keys after the lane's query row; a tile the
        // wave's first row already covers entirely needs no look
        if (kv0 + kBKV > N || (CAUSAL && kv0 + kBKV - 1 > q0 + 32 * wave_u)) {
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int key = kv0 + 32 * kb + (i & 3) + 8 * (i >> 2) + 4 * h;
                    if (key >= N || (CAUSAL && key > qrow)) s[kb][i] = -INFINITY;
                }
        }
        // row max: two independent chains of v_max3_f32 written out as single instructions (fmaxf
        // would first quiet a possible signalling NaN in each chain head), then the other half's
        float a = vmax3(s[0][0], s[0][1], s[0][2]), b = vmax3(s[1][0], s[1][1], s[1][2]);
#pragma unroll
        for (int i = 3; i < 15; i += 2) { a = vmax3(a, s[0][i], s[0][i + 1]); b = vmax3(b, s[1][i], s[1][i + 1]); }
        const float tloc = vmax3(a, b, vmax2(s[0][15], s[1][15]));
        if constexpr (STAMP) s1 = stamp();
        // a row's max grows past the threshold exactly when one of its two lanes' partial maxima
        // does: the lanes vote on their own 32 logits, the half exchange happens only in the (rare)
        // tile where the vote says the running max has to move
        if (__builtin_amdgcn_ballot_w64(__builtin_fmaf(tloc, sl2, -m) > thr) != 0) {   // wave-uniform
            const HalfPair hp = both_halves(tloc);
            const float mt = vmax2(hp.lo, hp.hi) * sl2;
            const float m_new = fmaxf(m, mt);
            const float alpha = __builtin_amdgcn_exp2f(m - m_new);
            m = m_new;
            l *= alpha;
#pragma unroll
            for (int db = 0; db < DB; ++db)
#pragma unroll
                for (int i = 0; i < 16; ++i) o[db][i] *= alpha;
        }
        if constexpr (STAMP) { s2 = stamp(); cyc_x2 += s2 - s1; }
        float psum = 0.f;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                s[kb][i] = __builtin_amdgcn_exp2f(__builtin_fmaf(s[kb][i], sl2, -m));
                psum += s[kb][i];
            }
        l += psum;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int st = 0; st < 2; ++st)
#pragma unroll
                for (int j = 0; j < 8; ++j) pb[2 * kb + st][j] = (T)s[kb][8 * st + j];
    };

    // key tiles of this workgroup (causal: up to the block's last row; the first tile always
    // holds key 0, which every row may see, so no row is ever fully masked there)
    const int T_ = CAUSAL ? min((N + kBKV - 1) / kBKV, (q0 + kBQ) / kBKV) : (N + kBKV - 1) / kBKV;

    // Own LDS-DMA fetches have landed / phase boundary without draining the ones in flight.
    auto dma_wait = [&]() __attribute__((always_inline)) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); };
    auto phase_barrier = [&]() __attribute__((always_inline)) {
        asm volatile("" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };

    // ---- prologue: K(0), V(0), K(1) into their ring slots, and this wave's 32 x D block of Q
    // into a private half slot (K slots 2, 3 and V slots 1, 2 are free until the first
    // barrier), laid out like the first 32 rows of a K tile.  Fetched this way the block is
    // read from memory as whole 1-KiB pieces; loading the fragments straight from global
    // memory (16 bytes per lane at a 2D-byte stride) took 3000-6000 cycles per workgroup in
    // the address pipeline, ahead of the K/V fetches queued behind it.
    // Every fetch in this kernel is unconditional (tiles past the end read zeros into a slot
    // nobody reads again).
    fetch_k(0, 0);
    fetch_v(0, 0);
    fetch_k(1, 1);
    const int qbyte = (wave_u < 2 ? 2 : wave_u < 4 ? 3 : wave_u < 6 ? 5 : 6) * TILE_B + (wave_u & 1) * (TILE_B / 2);
    {
        [[maybe_unused]] const __amdgpu_buffer_rsrc_t qrs = __builtin_amdgcn_make_buffer_rsrc((void*)(Qg + hoff), 0, head_bytes, 0x00020000);
        [[maybe_unused]] const int qrow0_b = (q0 + 32 * wave) * gd * 2;   // in the lane offset: it is bounds-checked
#pragma unroll
        for (int pc = 0; pc < TILE_B / 2 / 1024; ++pc) {
            const int u = 64 * pc + lane;
            const int grp8 = u / (SUB_B / 16), within = u % (SUB_B / 16);
            const int row = 8 * grp8 + ((within & 31) >> 2);
            [[maybe_unused]] const int ch = 4 * (within >> 5) + ((within & 3) ^ ((row >> 2) & 3));
#if defined(__HIP_DEVICE_COMPILE__)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(qrs, (lds_void*)(lds + qbyte + 1024 * pc), 16,
                                                     ch * 8 < gd ? qrow0_b + (row * gd + ch * 8) * 2 : kPastEndV1, 0, 0, 0);
#endif
        }
    }
    unsigned long long t_p1 = 0, t_p2 = 0, t_p3 = 0, t_p4 = 0;
    if constexpr (STAMP) t_p1 = stamp();
    dma_wait();
    if constexpr (STAMP) t_p2 = stamp();
    // Q fragments (row r, chunks 2ks + h) out of the private block; the reads must have
    // returned before the barrier, behind which K(2) and V(1) overwrite the Q blocks.
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
        qf[ks] = *reinterpret_cast<const vec8*>(lds + (kbase[ks & 1] + qbyte) + 512 * (ks >> 1));
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();
    if constexpr (STAMP) t_p3 = stamp();

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
    if constexpr (STAMP) t_p4 = stamp();
    if (grp) __syncthreads();
    softmax(0);
    dma_wait();                              // K(2), V(1): every wave's share is in before the barrier
    prefill(1, 0, IC<0>{}, IC<NQ + NP>{});   // first fragments of M(0): K(1), complete since the prologue
    phase_barrier();

    // Ring bookkeeping (wave-uniform): tile u of K lives in slot u % 4, of V in slot u % 3.
    // The fetches of K(w+3) and V(w+2) are issued under the last MFMAs of M(w), after the
    // last fragment request of that phase (an LDS-DMA issue costs the issuing wave ~100
    // cycles in a vector phase and next to nothing in the shadow of an MFMA).  Their slots
    // held K(w-1) and V(w-1), last read by M(w-2) / M(w-1), which for either half ended at
    // least one barrier before.  Every wave waits for its own fetches (dma_wait) at the end
    // of its X(w+1); the first readers are the prefill at the end of X(w+2) (K) and M(w+2)
    // (V), which for the leading half come one barrier after the lagging half's X(w+1).
    unsigned long long cyc_m = 0, cyc_mw = 0, cyc_x = 0, cyc_xw = 0, cyc_mm = 0;
    int k1 = 1, v0 = 0;              // (t+1) % 4, t % 3
    if constexpr (STAMP) t_loop0 = stamp();
    for (int t = 0; t + 1 < T_; ++t) {
        unsigned long long ta = 0, tb = 0, tc = 0, td = 0, te = 0;
        if constexpr (STAMP) ta = stamp();
        const int k0 = (k1 + 3) & 3;             // t % 4
        const int k2 = (k1 + 1) & 3;             // (t+2) % 4
        const int k3 = (k1 + 2) & 3;             // (t+3) % 4
        const int v1 = v0 == 2 ? 0 : v0 + 1;     // (t+1) % 3
        const int v2 = v1 == 2 ? 0 : v1 + 1;     // (t+2) % 3
        (void)k0;
        // ---- M(t): S(t+1) = K(t+1) Q^T ; O += V(t)^T P(t)^T ; fetch K(t+3), V(t+2)
        mma_stream(k1, v0, IC<0>{}, IC<NQ + NP>{}, [&](int i) __attribute__((always_inline)) {
            constexpr int L0 = NQ + NP - FW;     // no fragment requests from here on
#pragma unroll
            for (int j = 0; j < LPT; ++j) {
                if (i == L0 + 2 * j)     fetch_piece(krs, t + 3, k3 * TILE_B, j);
                if (i == L0 + 2 * j + 1) fetch_piece(vrs, t + 2, (4 + v2) * TILE_B, j);
            }
        });
        if constexpr (STAMP) tb = stamp();
        phase_barrier();                         // the fetches just issued stay in flight across it
        if constexpr (STAMP) tc = stamp();
        // ---- X(t+1)
        softmax((t + 1) * kBKV);
        dma_wait();                              // this wave's share of K(t+3), V(t+2) is in LDS
        prefill(k2, v1, IC<0>{}, IC<NQ + NP>{}); // first fragments of M(t+1): K(t+2)
        if constexpr (STAMP) td = stamp();
        phase_barrier();
        if constexpr (STAMP) {
            te = stamp();
            cyc_mm += tb - ta; cyc_m += tb - ta; cyc_mw += tc - tb;
            cyc_x += td - tc; cyc_xw += te - td;
        }
        k1 = k2; v0 = v1;
    }
    if constexpr (STAMP) t_loop1 = stamp();
    prefill(0, v0, IC<NQ>{}, IC<NQ + NP>{});
    mma_stream(0, v0, IC<NQ>{}, IC<NQ + NP>{}, nothing);   // M(T-1): O += V(T-1)^T P^T
    dma_wait();
    if (grp == 0) __syncthreads();   // the leading half meets the lagging half's last barrier

    const float ltot = half_exchange_sum(l);
    const float inv = 1.0f / ltot;
    {
        // O leaves through LDS so that it is stored as whole rows: each wave parks its 32 x D
        // tile in its own region (rows padded by 16 bytes against bank conflicts), then 64/CPR
        // rows go out per store instruction, 16 bytes per lane, each row contiguous.
        constexpr int OROW_B = D * 2 + 16;
        __syncthreads();                         // every wave is done with the K/V rings
        char* obuf = lds + wave * (32 * OROW_B);
#pragma unroll
        for (int db = 0; db < DB; ++db)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                vec4 w;
#pragma unroll
                for (int j = 0; j < 4; ++j) w[j] = (T)(o[db][4 * g + j] * inv);
                *reinterpret_cast<vec4*>(obuf + r * OROW_B + (32 * db + 8 * g + 4 * h) * 2) = w;
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
        if (LSEg != nullptr && h == 0 && qrow < N)
            LSEg[(size_t)head * N + qrow] = (m + __builtin_amdgcn_logf(ltot)) * 0.6931471805599453f;
    }
    if constexpr (STAMP) {
        if (dbg != nullptr && (wave == 0 || wave == 4) && lane == 0) {
            unsigned long long* d = dbg + (size_t)blockIdx.x * 24 + (wave >> 2) * 12;
            d[0] = cyc_m; d[1] = cyc_mw; d[2] = cyc_x; d[3] = cyc_xw; d[4] = cyc_mm; d[5] = t_loop0 - t_kernel0; d[6] = stamp() - t_loop1;
            d[7] = t_p1 - t_kernel0; d[8] = t_p2 - t_p1; d[9] = t_p3 - t_p2; d[10] = t_p4 - t_p3; d[11] = t_loop0 - t_p4;
        }
    }

}

template <typename T, int D>
int launch(const void* q, const void* k, const void* v, void* o, float* lse,
           int64_t heads, int64_t N, int64_t d, float scale, bool causal, hipStream_t s) {
    if (N * d * 2 >= (int64_t)kPastEndV1) return fail(MYFA_E_UNSUPPORTED, "attention: N*d=%lld per head too large", (long long)(N * d));
    const int64_t nqblk = cdiv(N, kBQ);
    const int64_t nwg = heads * nqblk;
    if (nwg > 0x7fffffffLL) return fail(MYFA_E_UNSUPPORTED, "attention: too many workgroups (%lld)", (long long)nwg);
    const float sl2 = scale * 1.4426950408889634f;
    const Tuning& tn = tuning();
    if (tn.attn_variant == 0 && !causal && d == D)
        launch_kernel((attn_fwd_mfma_kernel<T, D>), dim3((unsigned)nwg), dim3(kThreads), 0, s,
                           static_cast<const T*>(q), static_cast<const T*>(k), static_cast<const T*>(v),
                           static_cast<T*>(o), lse, (int)N, (int)nqblk, (int)nwg, sl2);
    else {
        auto go = [&](auto kern) {
            launch_kernel(kern, dim3((unsigned)nwg), dim3(kThreads), 0, s,
                               static_cast<const T*>(q), static_cast<const T*>(k), static_cast<const T*>(v),
                               static_cast<T*>(o), lse, (int)N, (int)nqblk, (int)nwg, sl2, tn.attn_thr < 0.f ? 8.f : tn.attn_thr, (int)d, tn.attn_dbg);
        };
#ifdef MYFA_TUNING
        if (causal && tn.attn_dbg != nullptr) go(attn_fwd_mfma_pp_kernel<T, D, true, true>);
        else if (tn.attn_dbg != nullptr) go(attn_fwd_mfma_pp_kernel<T, D, true>);
        else
#endif
        if (causal) go(attn_fwd_mfma_pp_kernel<T, D, false, true>);
        else go(attn_fwd_mfma_pp_kernel<T, D, false>);
    }
    return check_launch("attn_fwd_mfma");
}

}  // namespace

// The 32x32x16 builds (tuning flavour: attn_variant 0 and 1).  dtype: MYFA_F16 or MYFA_BF16;
// d: a multiple of 8 in 8..128 (run on the next build up, the missing columns read as zeros).
int attn_fwd_mfma32(const void* q, const void* k, const void* v, void* o, float* lse,
                    int64_t heads, int64_t N, int64_t d, float scale, int dtype, bool causal, hipStream_t s) {
    if (d >= 8 && d <= 128 && d % 8 == 0) {
        const bool big = d > 64;
        if (dtype == MYFA_BF16) return big ? launch<__bf16, 128>(q, k, v, o, lse, heads, N, d, scale, causal, s)
                                           : launch<__bf16, 64>(q, k, v, o, lse, heads, N, d, scale, causal, s);
        if (dtype == MYFA_F16)  return big ? launch<_Float16, 128>(q, k, v, o, lse, heads, N, d, scale, causal, s)
                                           : launch<_Float16, 64>(q, k, v, o, lse, heads, N, d, scale, causal, s);
    }
    return fail(MYFA_E_UNSUPPORTED, "attention (f16/bf16): head dim %lld is not a multiple of 8 in 8..128", (long long)d);
}

}  // namespace myfa
#endif  // MYFA_TUNING
