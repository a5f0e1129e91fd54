// Row-major float32 matrix product C[m,n] = A[m,k] B[k,n] on the matrix cores.
//
// Stands in for the reference's own tiled kernel matmul_kernel (csrc/my_flash_attn/matmul.cu:22-50:
// 16 x 16 shared-memory tiles, one output per thread) behind my_matmul / my_matmul_out
// (csrc/api.cpp:13-66).  The vendor-GEMM flavour my_matmul_cublas (csrc/api.cpp:68-105) stays
// with the platform's vendor GEMM in the Python layer.
//
// Workgroup = 4 waves = a 64 x 64 block of C, wave = 32 x 32 = 2 x 2 tiles of
// v_mfma_f32_16x16x4_f32 (lane (j = l & 15, g = l >> 4): a = A[j][g], b = B[g][j],
// result register r = C[4g + r][j]).  The k dimension is walked 16 at a time through an
// LDS copy of the A block (64 x 16, rows padded to 17) and the B block (16 x 64, rows padded
// to 68); the next blocks are fetched into registers while the current ones are multiplied.
// Any m, k, n: edges are zero-filled on the way in and skipped on the way out.
#include "myfa_common.h"

namespace myfa {
namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kBM = 64, kBN = 64, kBK = 16;
constexpr int kAS = kBK + 1;     // A block row stride in LDS (floats)
constexpr int kBS = kBN + 4;     // B block row stride in LDS (floats)

// VEC: k % 4 == 0, n % 4 == 0 and 16-byte aligned bases, so interior pieces move as float4.
template <bool VEC>
__global__ void __launch_bounds__(256)
sgemm_mfma_kernel(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C,
                  int m, int k, int n) {
    __shared__ float As[kBM * kAS];
    __shared__ float Bs[kBK * kBS];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 15, g = lane >> 4;
    const int m0 = blockIdx.y * kBM, n0 = blockIdx.x * kBN;
    const int wm = (wave >> 1) * 32, wn = (wave & 1) * 32;       // the wave's 32 x 32 corner in the block

    // staging: thread t moves A[m0 + t/4][k0 + 4(t%4) ..+3] and B[k0 + t/16][n0 + 4(t%16) ..+3]
    const int ar = tid >> 2, ac = (tid & 3) * 4;
    const int br = tid >> 4, bc = (tid & 15) * 4;
    f32x4 ra, rb;
    auto fetch = [&](int k0) __attribute__((always_inline)) {
        ra = f32x4{0.f, 0.f, 0.f, 0.f};
        rb = f32x4{0.f, 0.f, 0.f, 0.f};
        const int am = m0 + ar, ak = k0 + ac;
        if (am < m) {
            const float* p = A + (size_t)am * k + ak;
            if (VEC && ak + 3 < k) ra = *reinterpret_cast<const f32x4*>(p);
            else {
#pragma unroll
                for (int t = 0; t < 4; ++t)
                    if (ak + t < k) ra[t] = p[t];
            }
        }
        const int bk = k0 + br, bn = n0 + bc;
        if (bk < k) {
            const float* p = B + (size_t)bk * n + bn;
            if (VEC && bn + 3 < n) rb = *reinterpret_cast<const f32x4*>(p);
            else {
#pragma unroll
                for (int t = 0; t < 4; ++t)
                    if (bn + t < n) rb[t] = p[t];
            }
        }
    };
    auto publish = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int t = 0; t < 4; ++t) As[ar * kAS + ac + t] = ra[t];
        *reinterpret_cast<f32x4*>(&Bs[br * kBS + bc]) = rb;
    };

    f32x4 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

    fetch(0);
    for (int k0 = 0; k0 < k; k0 += kBK) {
        publish();
        __syncthreads();
        fetch(k0 + kBK);                       // (past the end: zero-filled, never published)
#pragma unroll
        for (int ks = 0; ks < kBK / 4; ++ks) {
            float af[2], bf[2];
#pragma unroll
            for (int a = 0; a < 2; ++a) af[a] = As[(wm + 16 * a + j) * kAS + 4 * ks + g];
#pragma unroll
            for (int b = 0; b < 2; ++b) bf[b] = Bs[(4 * ks + g) * kBS + wn + 16 * b + j];
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[a], bf[b], acc[a][b], 0, 0, 0);
        }
        __syncthreads();
    }

    // result register r of tile (a, b) is C[m0 + wm + 16a + 4g + r][n0 + wn + 16b + j]
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = m0 + wm + 16 * a + 4 * g + r, col = n0 + wn + 16 * b + j;
                if (row < m && col < n) C[(size_t)row * n + col] = acc[a][b][r];
            }
}

}  // namespace

int matmul_f32(const float* a, const float* b, float* c, int64_t m, int64_t k, int64_t n, hipStream_t s) {
    if (m < 0 || k < 0 || n < 0) return fail(MYFA_E_INVALID, "matmul: negative size m=%lld k=%lld n=%lld", (long long)m, (long long)k, (long long)n);
    if (m == 0 || n == 0) return MYFA_OK;
    if (!c || ((!a || !b) && k > 0)) return fail(MYFA_E_INVALID, "matmul: null pointer");
    if (m > 0x7fffffffLL || k > 0x7fffffffLL || n > 0x7fffffffLL || cdiv(m, kBM) > 65535)
        return fail(MYFA_E_UNSUPPORTED, "matmul: size beyond the kernel's limits");
    const dim3 grid((unsigned)cdiv(n, kBN), (unsigned)cdiv(m, kBM));
    const bool vec = k % 4 == 0 && n % 4 == 0 &&
                     ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b)) % 16 == 0);
    if (vec) hipLaunchKernelGGL(sgemm_mfma_kernel<true>, grid, dim3(256), 0, s, a, b, c, (int)m, (int)k, (int)n);
    else     hipLaunchKernelGGL(sgemm_mfma_kernel<false>, grid, dim3(256), 0, s, a, b, c, (int)m, (int)k, (int)n);
    return check_launch("matmul_f32");
}

}  // namespace myfa
