// Fused attention forward in float32 for any sequence length and head dim.
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
// oracle, the logits are accumulated in float64 (v_fma_f64) and the subtraction
// of the row max is done in float64; the exponential, the row sum and the P.V
// accumulation are float32.
//
// Mapping.  One lane owns one query row ("query on the lane"), so row max / row
// sum / rescale are per-lane scalars with no cross-lane traffic.  K and V
// elements are the same for every lane of a wave, so they are fetched with
// wave-uniform addresses (scalar loads, 16 dwords at a time) and fed to the FMAs
// as scalar operands; Q^T lives in LDS, padded to 65 floats per row so that both
// its fill and its per-lane reads are bank-conflict-free.  A workgroup is 4 waves
// working on the SAME 64 queries, each taking every 4th tile of 16 keys; the four
// partial (m, l, O) states are merged through LDS at the end.  Output columns are
// processed in chunks of DCH per workgroup (gridDim.z chunks).
#include "myfa_common.h"

namespace myfa {
namespace {

constexpr int kQB = 64;    // queries per workgroup (one per lane)
constexpr int kKT = 16;    // keys per tile
constexpr int kNW = 4;     // waves per workgroup
constexpr int kQPad = 65;  // padded row length of the Q^T image

__device__ __forceinline__ float exp_or_zero(float m_old, float m_new) {
    // exp(m_old - m_new) with the convention exp(-inf - (-inf)) = 0
    return m_old == -INFINITY ? 0.f : expf(m_old - m_new);
}

template <int DCH, bool KCONTIG>
__global__ void __launch_bounds__(kNW * 64)
attn_fwd_f32_kernel(const float* __restrict__ Qg, const float* __restrict__ Kg,
                    const float* __restrict__ Vg, float* __restrict__ Og, float* __restrict__ LSEg,
                    int N, int d, long long k_key_stride, long long k_d_stride,
                    long long head_stride, float scale) {
    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int q0 = blockIdx.x * kQB;
    const int c0 = blockIdx.z * DCH;
    const long long hoff = (long long)blockIdx.y * head_stride;
    const float* Qh = Qg + hoff;
    const float* Kh = Kg + hoff;
    const float* Vh = Vg + hoff;

    // ---- Q^T image: smem[c * 65 + row]; rows past N repeat the last valid row
    const int nq = min(kQB, N - q0);
    for (int idx = threadIdx.x; idx < kQB * d; idx += kNW * 64) {
        const int row = idx / d, c = idx - row * d;
        const int grow = q0 + (row < nq ? row : nq - 1);
        smem[c * kQPad + row] = Qh[(long long)grow * d + c];
    }
    __syncthreads();

    float m = -INFINITY, l = 0.f;
    float acc[DCH];
#pragma unroll
    for (int i = 0; i < DCH; ++i) acc[i] = 0.f;

    const bool dfull = (c0 + DCH <= d);
    const double dscale = (double)scale;
    const int ntiles = (N + kKT - 1) / kKT;

    for (int t = wave; t < ntiles; t += kNW) {
        const int j0 = t * kKT;
        const bool full = (j0 + kKT <= N);

        // ---- logits of 16 keys for this lane's query, float64 accumulation
        double s[kKT];
#pragma unroll
        for (int j = 0; j < kKT; ++j) s[j] = 0.0;
        if (full) {
            for (int c = 0; c < d; ++c) {
                const double qv = (double)smem[c * kQPad + lane];
                const float* kp = Kh + (long long)c * k_d_stride + (KCONTIG ? (long long)j0 : (long long)j0 * k_key_stride);
#pragma unroll
                for (int j = 0; j < kKT; ++j) {
                    const float kv = KCONTIG ? kp[j] : kp[(long long)j * k_key_stride];
                    s[j] = fma(qv, (double)kv, s[j]);
                }
            }
        } else {
            for (int c = 0; c < d; ++c) {
                const double qv = (double)smem[c * kQPad + lane];
                const float* kp = Kh + (long long)c * k_d_stride;
#pragma unroll
                for (int j = 0; j < kKT; ++j) {
                    const int jj = min(j0 + j, N - 1);
                    const float kv = KCONTIG ? kp[jj] : kp[(long long)jj * k_key_stride];
                    s[j] = fma(qv, (double)kv, s[j]);
                }
            }
        }

        double tmax = -INFINITY;
#pragma unroll
        for (int j = 0; j < kKT; ++j) {
            s[j] = (j0 + j < N) ? s[j] * dscale : -INFINITY;
            tmax = fmax(tmax, s[j]);
        }
        const float m_new = fmaxf(m, (float)tmax);
        const float alpha = exp_or_zero(m, m_new);
        float p[kKT];
        float psum = 0.f;
#pragma unroll
        for (int j = 0; j < kKT; ++j) {
            p[j] = expf((float)(s[j] - (double)m_new));
            psum += p[j];
        }
        l = l * alpha + psum;
        m = m_new;
#pragma unroll
        for (int i = 0; i < DCH; ++i) acc[i] *= alpha;

        // ---- O[query, c0..c0+DCH) += sum_j p_j V[j, ...]
        if (dfull) {
#pragma unroll
            for (int j = 0; j < kKT; ++j) {
                const float* vp = Vh + (long long)min(j0 + j, N - 1) * d + c0;
#pragma unroll
                for (int i = 0; i < DCH; ++i) acc[i] = fmaf(p[j], vp[i], acc[i]);
            }
        } else {
#pragma unroll
            for (int j = 0; j < kKT; ++j) {
                const float* vp = Vh + (long long)min(j0 + j, N - 1) * d;
#pragma unroll
                for (int i = 0; i < DCH; ++i) acc[i] = fmaf(p[j], vp[min(c0 + i, d - 1)], acc[i]);
            }
        }
    }

    // ---- merge the four waves' partial states: (1 -> 0, 3 -> 2), then (2 -> 0)
    __syncthreads();   // Q^T image no longer needed; its space becomes the merge buffers
    constexpr int kSlot = (DCH + 2) * 64;
    auto put = [&](float* b) {
        b[lane] = m;
        b[64 + lane] = l;
#pragma unroll
        for (int i = 0; i < DCH; ++i) b[(2 + i) * 64 + lane] = acc[i];
    };
    auto merge = [&](const float* b) {
        const float m2 = b[lane], l2 = b[64 + lane];
        const float mn = fmaxf(m, m2);
        const float a1 = exp_or_zero(m, mn), a2 = exp_or_zero(m2, mn);
        l = l * a1 + l2 * a2;
#pragma unroll
        for (int i = 0; i < DCH; ++i) acc[i] = acc[i] * a1 + b[(2 + i) * 64 + lane] * a2;
        m = mn;
    };
    if (wave == 1 || wave == 3) put(smem + (wave >> 1) * kSlot);
    __syncthreads();
    if (wave == 0 || wave == 2) merge(smem + (wave >> 1) * kSlot);
    __syncthreads();
    if (wave == 2) put(smem);
    __syncthreads();
    if (wave == 0) {
        merge(smem);
        const int row = q0 + lane;
        if (row < N) {
            const float inv = 1.0f / l;
            float* orow = Og + hoff + (long long)row * d;
#pragma unroll
            for (int i = 0; i < DCH; ++i)
                if (c0 + i < d) orow[c0 + i] = acc[i] * inv;
            if (LSEg != nullptr && blockIdx.z == 0)
                LSEg[(long long)blockIdx.y * N + row] = m + logf(l);
        }
    }
}

template <int DCH, bool KCONTIG>
int launch(const float* q, const float* k, const float* v, float* o, float* lse,
           int64_t heads, int64_t N, int64_t d, int64_t k_key_stride, int64_t k_d_stride,
           float scale, hipStream_t s) {
    auto kern = attn_fwd_f32_kernel<DCH, KCONTIG>;
    const size_t lds_floats = (size_t)max((int64_t)d * kQPad, (int64_t)2 * (DCH + 2) * 64);
    const size_t lds_bytes = lds_floats * sizeof(float);
    if (lds_bytes > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (e != hipSuccess)
            return fail(MYFA_E_LAUNCH, "attention (f32): cannot reserve %zu bytes of LDS: %s", lds_bytes, hipGetErrorString(e));
    }
    const dim3 grid((unsigned)cdiv(N, kQB), (unsigned)heads, (unsigned)cdiv(d, DCH));
    hipLaunchKernelGGL(kern, grid, dim3(kNW * 64), lds_bytes, s, q, k, v, o, lse, (int)N, (int)d,
                       (long long)k_key_stride, (long long)k_d_stride, (long long)(N * d), scale);
    return check_launch("attn_fwd_f32");
}

}  // namespace

// K element (key j, dim c) of a head is at k[j * k_key_stride + c * k_d_stride]:
//   K_t layout [d, N] (the reference's my_attention): k_key_stride = 1, k_d_stride = N
//   K   layout [N, d] (batched entry point):          k_key_stride = d, k_d_stride = 1
int attn_fwd_f32(const float* q, const float* k, const float* v, float* o, float* lse,
                 int64_t heads, int64_t N, int64_t d, int64_t k_key_stride, int64_t k_d_stride,
                 float scale, hipStream_t s) {
    if (d > 512) return fail(MYFA_E_UNSUPPORTED, "attention (f32): head dim %lld > 512", (long long)d);
    if (N > (1LL << 30) || heads > 65535)
        return fail(MYFA_E_UNSUPPORTED, "attention (f32): N=%lld or heads=%lld beyond the grid limits", (long long)N, (long long)heads);
    const bool kc = (k_key_stride == 1);
    if (d <= 32)
        return kc ? launch<32, true>(q, k, v, o, lse, heads, N, d, k_key_stride, k_d_stride, scale, s)
                  : launch<32, false>(q, k, v, o, lse, heads, N, d, k_key_stride, k_d_stride, scale, s);
    return kc ? launch<64, true>(q, k, v, o, lse, heads, N, d, k_key_stride, k_d_stride, scale, s)
              : launch<64, false>(q, k, v, o, lse, heads, N, d, k_key_stride, k_d_stride, scale, s);
}

}  // namespace myfa
