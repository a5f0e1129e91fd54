// Fused attention forward for f16 / bf16 with head dims above 128, up to 256.
//
//     O[b,h] = softmax(scale * Q[b,h] K[b,h]^T) V[b,h]      Q,K,V,O: [B,H,N,d], d a multiple of 8 in 136..256
//
// Same math and register choreography as attn_fwd_mfma.hip (the online-softmax recurrence of
// tests/test_flash_attn.py:82-93 on v_mfma_f32_32x32x16, S^T = K Q^T so that a query's logits sit in
// one lane pair, O^T = V^T P^T with the transposing LDS read); what differs is the budget.  At 256
// columns a wave's O^T tile alone is 128 accumulator registers and its Q fragments 64 more, so a
// wave needs the whole register file of its SIMD:
//   workgroup = 4 waves = 128 query rows, one wave per SIMD (512 registers), wave w owns rows 32w..32w+31;
//   K/V tiles of 64 keys x 256 columns (32 KiB each), double-buffered in LDS (128 KiB), staged
//   through registers one tile ahead (buffer loads: rows past N and columns past d read as zeros,
//   so any multiple of 8 up to 256 runs on this build), one barrier per tile;
//   matrix and vector work of a tile run in the one wave, in the compiler's interleaving.  A tile is
//   64 MFMAs (2048 cycles) against ~250 vector instructions, so the matrix pipe stays the larger
//   share without a second wave to overlap with.
// The running max moves only in a tile where some row's max grew by more than `thr` (log2 units;
// P may reach 2^thr), which keeps the 128-register rescale of O^T out of most tiles.
#include "attn_mfma_common.h"

namespace myfa {
namespace {

constexpr int kWavesW = 4;
constexpr int kThreadsW = kWavesW * 64;
constexpr int kBQW = kWavesW * 32;   // query rows per workgroup
constexpr int kBKVW = 64;            // keys per tile
constexpr int kDW = 256;             // columns of the build
// A buffer offset no tensor reaches (the launcher keeps a head below it) and that a tile offset
// added to it cannot wrap: fetches from there read zero without touching memory.
constexpr int kPastEndW = 0x7FFF0000;

// LDS images of a [64][256] tile, row-major with the 16-byte chunk index XOR-swizzled inside each
// half row: K for row reads (ds_read_b128, 32 rows x 2 chunks per instruction), V for the
// transposing reads (ds_read_b64_tr_b16).
__device__ __forceinline__ int k_off_w(int row, int ch) {
    return row * (2 * kDW) + (((ch & 16) | ((ch & 15) ^ (row & 15))) << 4);
}
__device__ __forceinline__ int v_off_w(int row, int ch) {
    return row * (2 * kDW) + (((ch & 16) | ((ch & 15) ^ (((row & 3) << 2) | ((row >> 2) & 3)))) << 4);
}

template <typename T, bool CAUSAL>
__global__ void __launch_bounds__(kThreadsW, 1)
attn_fwd_mfma_wide_kernel(const T* __restrict__ Qg, const T* __restrict__ Kg, const T* __restrict__ Vg,
                          T* __restrict__ Og, float* __restrict__ LSEg, int N, int nqblk, int nwg,
                          float sl2, float thr, int gd) {
    using M = Mma<T>;
    using vec8 = typename M::vec8;
    using vec4 = typename M::vec4;
    constexpr int D = kDW;
    constexpr int CPR = D / 8;                    // 16-byte chunks per row
    constexpr int TILE_B = kBKVW * D * 2;         // bytes per K (or V) tile
    constexpr int KS = D / 16;                    // k-steps of S^T = K Q^T
    constexpr int DB = D / 32;                    // 32-row blocks of O^T
    constexpr int LPT = kBKVW * CPR / kThreadsW;  // 16-byte loads per thread per tile

    __shared__ __attribute__((aligned(16))) char lds[4 * TILE_B];   // K0 K1 V0 V1

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

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;

    // gd <= D is the head dim in memory: rows are gd elements apart, chunks past it read as zeros
    const size_t hoff = (size_t)head * (size_t)N * gd;
    const unsigned head_bytes = (unsigned)N * (unsigned)gd * 2u;
    const __amdgpu_buffer_rsrc_t qrs = __builtin_amdgcn_make_buffer_rsrc((void*)(Qg + hoff), 0, head_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t krs = __builtin_amdgcn_make_buffer_rsrc((void*)(Kg + hoff), 0, head_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t vrs = __builtin_amdgcn_make_buffer_rsrc((void*)(Vg + hoff), 0, head_bytes, 0x00020000);

    // ---- Q^T fragments: lane (r, h) holds Q[row][16ks + 8h .. +7] for ks = 0..KS-1 (rows past N: zeros)
    const int qrow = q0 + wave * 32 + r;
    vec8 qf[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        const int ch = 2 * ks + h;
        const int off = ch * 8 < gd ? qrow * gd * 2 + ch * 16 : kPastEndW;
        qf[ks] = __builtin_bit_cast(vec8, __builtin_amdgcn_raw_buffer_load_b128(qrs, off, 0, 0));
    }

    // ---- staging of one K/V tile through registers: thread -> chunk (tid & 31) of rows (tid >> 5) + 8i
    const int st_row = tid >> 5, st_ch = tid & 31;
    const int st_off = st_ch * 8 < gd ? st_row * gd * 2 + st_ch * 16 : kPastEndW;
    u32x4 kst[LPT], vst[LPT];
    auto stage_load = [&](int kv0) {
        const int soff = kv0 * gd * 2;            // (past the last tile: past the end of the head, zeros)
#pragma unroll
        for (int i = 0; i < LPT; ++i) {
            kst[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(krs, st_off + 8 * i * gd * 2, soff, 0));
            vst[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(vrs, st_off + 8 * i * gd * 2, soff, 0));
        }
    };
    auto stage_write = [&](int buf) {
#pragma unroll
        for (int i = 0; i < LPT; ++i) {
            const int row = st_row + 8 * i;
            *reinterpret_cast<u32x4*>(lds + buf * TILE_B + k_off_w(row, st_ch)) = kst[i];
            *reinterpret_cast<u32x4*>(lds + (2 + buf) * TILE_B + v_off_w(row, st_ch)) = vst[i];
        }
    };

    f32x16 o[DB];
#pragma unroll
    for (int db = 0; db < DB; ++db)
#pragma unroll
        for (int i = 0; i < 16; ++i) o[db][i] = 0.f;
    float m = -INFINITY;   // running (deferred) row max of scale*log2(e)*S
    float l = 0.f;         // this lane half's share of the running row sum

    const int ntiles_all = (N + kBKVW - 1) / kBKVW;
    const int ntiles = CAUSAL ? min(ntiles_all, (q0 + kBQW) / kBKVW) : ntiles_all;
    // the last key a row may see: tile 0 always holds key 0, which every row sees
    const int key_limit = CAUSAL ? min(N - 1, qrow) : N - 1;
    stage_load(0);
    stage_write(0);
    __syncthreads();

    // lane-constant parts of the transposing-read address
    const int tq = (lane & 15) >> 2, tp = lane & 3, tg1 = (lane >> 4) & 1;

    for (int t = 0; t < ntiles; ++t) {
        const int buf = t & 1;
        const int kv0 = t * kBKVW;
        stage_load(kv0 + kBKVW);

        const char* Kb = lds + buf * TILE_B;
        const char* Vb = lds + (2 + buf) * TILE_B;

        // ---- S^T = K Q^T : 2 key blocks x KS k-steps.  One wave per SIMD has nobody to hide an LDS
        // read's latency behind, so fragments are read eight at a time, a batch ahead of the MFMAs
        // that consume them.
        f32x16 s[2];
        constexpr int KB = 8;                     // fragments per batch
        s16x8 ka[2][KB];
#pragma unroll
        for (int j = 0; j < KB; ++j) ka[0][j] = *reinterpret_cast<const s16x8*>(Kb + k_off_w(r, 2 * j + h));
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
            for (int i = 0; i < 16; ++i) s[kb][i] = 0.f;
#pragma unroll
            for (int half = 0; half < KS / KB; ++half) {
                const int cur = (kb * (KS / KB) + half) & 1, nb = kb * (KS / KB) + half + 1;    // next batch: (key block, half)
                if (nb < 2 * (KS / KB)) {
#pragma unroll
                    for (int j = 0; j < KB; ++j)
                        ka[cur ^ 1][j] = *reinterpret_cast<const s16x8*>(Kb + k_off_w(32 * (nb / (KS / KB)) + r, 2 * (KB * (nb % (KS / KB)) + j) + h));
                }
#pragma unroll
                for (int j = 0; j < KB; ++j) s[kb] = M::run(ka[cur][j], qf[KB * half + j], s[kb]);
            }
        }

        // ---- keys past N (last tile) and, under the causal mask, keys after the row (uniform test per wave)
        if (kv0 + kBKVW > N || (CAUSAL && kv0 + kBKVW - 1 > q0 + wave * 32)) {
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int key = kv0 + 32 * kb + (i & 3) + 8 * (i >> 2) + 4 * h;
                    if (key > key_limit) s[kb][i] = -INFINITY;
                }
        }

        // ---- online softmax (per query = per lane, two halves cooperate)
        float c0 = vmax3(s[0][0], s[0][1], s[0][2]), c1 = vmax3(s[1][0], s[1][1], s[1][2]);
#pragma unroll
        for (int i = 3; i + 1 < 16; i += 2) {
            c0 = vmax3(c0, s[0][i], s[0][i + 1]);
            c1 = vmax3(c1, s[1][i], s[1][i + 1]);
        }
        const float tmax = half_exchange_max(vmax3(c0, c1, vmax2(s[0][15], s[1][15])));
        const float mt = tmax * sl2;
        if (__builtin_amdgcn_ballot_w64(mt - m > thr) != 0) {   // wave-uniform; always taken in tile 0 (m = -inf)
            const float m_new = fmaxf(m, mt);
            const float alpha = __builtin_amdgcn_exp2f(m - m_new);
            m = m_new;
            l *= alpha;
#pragma unroll
            for (int db = 0; db < DB; ++db)
#pragma unroll
                for (int i = 0; i < 16; ++i) o[db][i] *= alpha;
        }
        float psum = 0.f;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                s[kb][i] = __builtin_amdgcn_exp2f(__builtin_fmaf(s[kb][i], sl2, -m));
                psum += s[kb][i];
            }
        l += psum;

        // ---- O^T += V^T P^T : 4 k-steps of 16 keys x DB row blocks, the V^T fragments of a k-step read
        // together, one k-step ahead
        auto v_frag = [&](int step, int db) {
            const int r0 = 16 * step + 4 * h + tq;             // key row of the first 4-row block
            const int ch = 4 * db + 2 * tg1 + (tp >> 1);
            const s16x4 lo = lds_read_tr(Vb + v_off_w(r0, ch) + 8 * (tp & 1));
            const s16x4 hi = lds_read_tr(Vb + v_off_w(r0 + 8, ch) + 8 * (tp & 1));
            s16x8 a;
            a[0] = lo[0]; a[1] = lo[1]; a[2] = lo[2]; a[3] = lo[3];
            a[4] = hi[0]; a[5] = hi[1]; a[6] = hi[2]; a[7] = hi[3];
            return a;
        };
        s16x8 va[2][DB];
#pragma unroll
        for (int db = 0; db < DB; ++db) va[0][db] = v_frag(0, db);
#pragma unroll
        for (int step = 0; step < 4; ++step) {
            if (step + 1 < 4) {
#pragma unroll
                for (int db = 0; db < DB; ++db) va[(step + 1) & 1][db] = v_frag(step + 1, db);
            }
            vec8 pb;
#pragma unroll
            for (int j = 0; j < 8; ++j) pb[j] = (T)s[step >> 1][8 * (step & 1) + j];
#pragma unroll
            for (int db = 0; db < DB; ++db) o[db] = M::run(va[step & 1][db], pb, o[db]);
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
        T* orow = Og + hoff + (size_t)qrow * gd;
#pragma unroll
        for (int db = 0; db < DB; ++db)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int col = 32 * db + 8 * g + 4 * h;
                if (col < gd) {               // (gd is a multiple of 8: the four columns are in or out together)
                    vec4 w;
#pragma unroll
                    for (int j = 0; j < 4; ++j) w[j] = (T)(o[db][4 * g + j] * inv);
                    *reinterpret_cast<vec4*>(orow + col) = w;
                }
            }
        if (LSEg != nullptr && h == 0)
            LSEg[(size_t)head * N + qrow] = (m + __builtin_amdgcn_logf(ltot)) * 0.6931471805599453f;
    }
}

template <typename T>
int launch_wide(const void* q, const void* k, const void* v, void* o, float* lse,
                int64_t heads, int64_t N, int64_t d, float scale, float thr, bool causal, hipStream_t s) {
    if (N * d * 2 >= (int64_t)kPastEndW) return fail(MYFA_E_UNSUPPORTED, "attention: N*d=%lld per head too large", (long long)(N * d));
    const int64_t nqblk = cdiv(N, (int64_t)kBQW);
    const int64_t nwg = heads * nqblk;
    if (nwg > 0x7fffffffLL) return fail(MYFA_E_UNSUPPORTED, "attention: too many workgroups (%lld)", (long long)nwg);
    const float sl2 = scale * 1.4426950408889634f;
    auto go = [&](auto kern) {
        launch_kernel(kern, dim3((unsigned)nwg), dim3(kThreadsW), 0, s,
                      static_cast<const T*>(q), static_cast<const T*>(k), static_cast<const T*>(v),
                      static_cast<T*>(o), lse, (int)N, (int)nqblk, (int)nwg, sl2, thr, (int)d);
    };
    if (causal) go(attn_fwd_mfma_wide_kernel<T, true>);
    else go(attn_fwd_mfma_wide_kernel<T, false>);
    return check_launch("attn_fwd_mfma_wide");
}

}  // namespace

// dtype: MYFA_F16 or MYFA_BF16; d: a multiple of 8 in 136..256; heads = B*H; scale > 0
int attn_fwd_mfma_wide(const void* q, const void* k, const void* v, void* o, float* lse,
                       int64_t heads, int64_t N, int64_t d, float scale, float thr, int dtype, bool causal, hipStream_t s) {
    if (d <= 128 || d > kDW || d % 8 != 0)
        return fail(MYFA_E_UNSUPPORTED, "attention (f16/bf16, wide): head dim %lld is not a multiple of 8 in 136..256", (long long)d);
    if (dtype == MYFA_BF16) return launch_wide<__bf16>(q, k, v, o, lse, heads, N, d, scale, thr, causal, s);
    if (dtype == MYFA_F16) return launch_wide<_Float16>(q, k, v, o, lse, heads, N, d, scale, thr, causal, s);
    return fail(MYFA_E_INVALID, "attention (f16/bf16, wide): dtype tag %d", dtype);
}

}  // namespace myfa
