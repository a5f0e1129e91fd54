// Row-major float32 matrix product C[m,n] = A[m,k] B[k,n] on the matrix cores.
//
// Stands in for the reference's own tiled kernel matmul_kernel (csrc/my_flash_attn/matmul.cu:22-50:
// 16 x 16 shared-memory tiles, one output per thread) behind my_matmul / my_matmul_out
// (csrc/api.cpp:13-66).  The vendor-GEMM flavour my_matmul_cublas (csrc/api.cpp:68-105) stays
// with the platform's vendor GEMM in the Python layer.
//
// Two kernels.  Small problems: workgroup = 4 waves = a 64 x 64 block of C, wave = 32 x 32 = 2 x 2 tiles of
// v_mfma_f32_16x16x4_f32 (lane (j = l & 15, g = l >> 4): a = A[j][g], b = B[g][j],
// result register r = C[4g + r][j]).  The k dimension is walked 16 at a time through an
// LDS copy of the A block (64 x 16, rows padded to 17) and the B block (16 x 64, rows padded
// to 68); the blocks of the next four chunks are on their way into registers while the current
// ones are multiplied.  Large problems (512 or more 128 x 128 blocks): sgemm_mfma_128_kernel
// below.  Any m, k, n: edges are zero-filled on the way in and skipped on the way out.
#include "myfa_common.h"
#include "tuning.h"


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
                  int m, int k, int n, int kslice) {
    // two copies of each block: chunk i + 1 is published while chunk i is multiplied, one barrier per chunk
    __shared__ float As[2][kBM * kAS];
    __shared__ __attribute__((aligned(16))) float Bs[2][kBK * kBS];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 15, g = lane >> 4;
    const int m0 = blockIdx.y * kBM, n0 = blockIdx.x * kBN;
    const int wm = (wave >> 1) * 32, wn = (wave & 1) * 32;       // the wave's 32 x 32 corner in the block
    // k split over gridDim.z workgroups (small problems: one 4-wave workgroup per CU cannot hide its
    // own load latency): this one sums k in [kbeg, kend) and adds its block to C, which the launcher
    // has zeroed, with float atomics.  Two addends per element, so the result does not depend on
    // which arrives first.  kslice is a multiple of kBK; rows of A stay k apart.
    const int kbeg = blockIdx.z * kslice, kend = min(k, kbeg + kslice);

    // staging: thread t moves A[m0 + t/4][k0 + 4(t%4) ..+3] and B[k0 + t/16][n0 + 4(t%16) ..+3]
    const int ar = tid >> 2, ac = (tid & 3) * 4;
    const int br = tid >> 4, bc = (tid & 15) * 4;
    // kPF chunks are in flight: with one 4-wave workgroup per CU (a 1024^3 problem is 256 blocks)
    // nothing else covers the latency of a load
    constexpr int kPF = 4;
    f32x4 ra[kPF], rb[kPF];
    // VEC: branch-free through buffer loads whose descriptors are rebased on the chunk (see the
    // 128 x 128 kernel below)
    auto fetch = [&](int k0, int slot) __attribute__((always_inline)) {
        if constexpr (VEC) {
#if defined(__HIP_DEVICE_COMPILE__)
            const int arows = min(kBM, m - m0), brows = min(kBK, kend - k0);
            const bool any = k0 < kend;
            const unsigned abytes = any ? (unsigned)(((size_t)(arows - 1) * k + (kend - k0)) * 4) : 0u;
            const unsigned bbytes = any ? (unsigned)(((size_t)(brows - 1) * n + (n - n0)) * 4) : 0u;
            const __amdgpu_buffer_rsrc_t ars = __builtin_amdgcn_make_buffer_rsrc((void*)(A + (size_t)m0 * k + (any ? k0 : 0)), 0, abytes, 0x00020000);
            const __amdgpu_buffer_rsrc_t brs = __builtin_amdgcn_make_buffer_rsrc((void*)(B + (size_t)(any ? k0 : 0) * n + n0), 0, bbytes, 0x00020000);
            const int aoff = k0 + ac < kend ? ac * 4 : (int)0x80000000;
            const int boff = n0 + bc < n ? bc * 4 : (int)0x80000000;
            ra[slot] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ars, aoff + ar * k * 4, 0, 0));
            rb[slot] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(brs, boff + br * n * 4, 0, 0));
#endif
        } else {
            ra[slot] = f32x4{0.f, 0.f, 0.f, 0.f};
            rb[slot] = f32x4{0.f, 0.f, 0.f, 0.f};
            const int am = m0 + ar, ak = k0 + ac;
            if (am < m) {
                const float* p = A + (size_t)am * k + ak;
#pragma unroll
                for (int t = 0; t < 4; ++t)
                    if (ak + t < kend) ra[slot][t] = p[t];
            }
            const int bk = k0 + br, bn = n0 + bc;
            if (bk < kend) {
                const float* p = B + (size_t)bk * n + bn;
#pragma unroll
                for (int t = 0; t < 4; ++t)
                    if (bn + t < n) rb[slot][t] = p[t];
            }
        }
    };
    auto publish = [&](int slot, int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int t = 0; t < 4; ++t) As[buf][ar * kAS + ac + t] = ra[slot][t];
        *reinterpret_cast<f32x4*>(&Bs[buf][br * kBS + bc]) = rb[slot];
    };

    f32x4 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

    static_assert(kPF % 2 == 0, "the LDS copy of a chunk is chosen by the parity of its slot");
#pragma unroll
    for (int p = 0; p < kPF; ++p) fetch(kbeg + p * kBK, p);      // (past the end: zero-filled, published, never read)
    publish(0, 0);
    fetch(kbeg + kPF * kBK, 0);
    __syncthreads();
    for (int kb = kbeg; kb < kend; kb += kPF * kBK) {
#pragma unroll
        for (int p = 0; p < kPF; ++p) {
            const int k0 = kb + p * kBK;
            if (k0 < kend) {                               // uniform over the workgroup
                // chunk k0 is in copy p & 1; the next chunk goes into the other copy (last read one
                // barrier ago) and its registers are refilled with the chunk kPF further on
                publish((p + 1) % kPF, (p + 1) & 1);
                fetch(k0 + (kPF + 1) * kBK, (p + 1) % kPF);
#pragma unroll
                for (int ks = 0; ks < kBK / 4; ++ks) {
                    float af[2], bf[2];
#pragma unroll
                    for (int a = 0; a < 2; ++a) af[a] = As[p & 1][(wm + 16 * a + j) * kAS + 4 * ks + g];
#pragma unroll
                    for (int b = 0; b < 2; ++b) bf[b] = Bs[p & 1][(4 * ks + g) * kBS + wn + 16 * b + j];
#pragma unroll
                    for (int a = 0; a < 2; ++a)
#pragma unroll
                        for (int b = 0; b < 2; ++b)
                            acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[a], bf[b], acc[a][b], 0, 0, 0);
                }
                __syncthreads();
            }
        }
    }

    // result register r of tile (a, b) is C[m0 + wm + 16a + 4g + r][n0 + wn + 16b + j]
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = m0 + wm + 16 * a + 4 * g + r, col = n0 + wn + 16 * b + j;
                if (row < m && col < n) {
                    if (gridDim.z > 1) __hip_atomic_fetch_add(&C[(size_t)row * n + col], acc[a][b][r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    else C[(size_t)row * n + col] = acc[a][b][r];
                }
            }
}

// The large-problem kernel: a 128 x 128 block of C per workgroup, wave = 64 x 64 = 4 x 4 tiles.
// Twice the block edge halves the bytes fetched per multiply-add (the 64 x 64 kernel asks L2 for
// 1 byte per 8 multiply-adds, about 10 TB/s at the MFMA peak).  The k index of the MFMA is
// permuted -- lane group g takes k = 4g + ks in step ks -- so that a lane's four A values of a
// 16-deep k block are adjacent in LDS and come as one 16-byte read.
constexpr int kLM = 128, kLN = 128;
constexpr int kLAS = kBK + 4;    // A block row stride in LDS (floats; 16-byte aligned rows)
constexpr int kLBS = kLN + 4;    // B block row stride in LDS (floats)

template <bool VEC>
__global__ void __launch_bounds__(256)
sgemm_mfma_128_kernel(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C,
                      int m, int k, int n) {
    // two copies of each block: chunk i + 1 is published while chunk i is multiplied, one barrier per chunk
    __shared__ __attribute__((aligned(16))) float As[2][kLM * kLAS];
    __shared__ __attribute__((aligned(16))) float Bs[2][kBK * kLBS];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 15, g = lane >> 4;
    const int m0 = blockIdx.y * kLM, n0 = blockIdx.x * kLN;
    const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;

    // staging: thread t moves A[m0 + t/4 (+64)][k0 + 4(t%4) ..+3] and B[k0 + t/32 (+8)][n0 + 4(t%32) ..+3]
    const int ar = tid >> 2, ac = (tid & 3) * 4;
    const int br = tid >> 5, bc = (tid & 31) * 4;
    f32x4 ra[2], rb[2];
    // VEC: branch-free through buffer loads.  The descriptors are rebased on the chunk every time
    // (base = first element the block reads of this chunk, size = up to the last valid element of
    // the block's last row), so rows past the matrix read as zero by the bounds check; a column
    // past the row's end would land in the next row and is sent out of range instead.
    auto fetch = [&](int k0) __attribute__((always_inline)) {
        if constexpr (VEC) {
#if defined(__HIP_DEVICE_COMPILE__)
            const int arows = min(kLM, m - m0), brows = min(kBK, k - k0);
            const bool any = k0 < k;
            const unsigned abytes = any ? (unsigned)(((size_t)(arows - 1) * k + (k - k0)) * 4) : 0u;
            const unsigned bbytes = any ? (unsigned)(((size_t)(brows - 1) * n + (n - n0)) * 4) : 0u;
            const __amdgpu_buffer_rsrc_t ars = __builtin_amdgcn_make_buffer_rsrc((void*)(A + (size_t)m0 * k + (any ? k0 : 0)), 0, abytes, 0x00020000);
            const __amdgpu_buffer_rsrc_t brs = __builtin_amdgcn_make_buffer_rsrc((void*)(B + (size_t)(any ? k0 : 0) * n + n0), 0, bbytes, 0x00020000);
            const int aoff = k0 + ac < k ? ac * 4 : (int)0x80000000;
            const int boff = n0 + bc < n ? bc * 4 : (int)0x80000000;
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                ra[u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ars, aoff + (ar + 64 * u) * k * 4, 0, 0));
                rb[u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(brs, boff + (br + 8 * u) * n * 4, 0, 0));
            }
#endif
        } else {
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                ra[u] = f32x4{0.f, 0.f, 0.f, 0.f};
                rb[u] = f32x4{0.f, 0.f, 0.f, 0.f};
                const int am = m0 + ar + 64 * u, ak = k0 + ac;
                if (am < m) {
                    const float* p = A + (size_t)am * k + ak;
#pragma unroll
                    for (int t = 0; t < 4; ++t)
                        if (ak + t < k) ra[u][t] = p[t];
                }
                const int bk = k0 + br + 8 * u, bn = n0 + bc;
                if (bk < k) {
                    const float* p = B + (size_t)bk * n + bn;
#pragma unroll
                    for (int t = 0; t < 4; ++t)
                        if (bn + t < n) rb[u][t] = p[t];
                }
            }
        }
    };
    auto publish = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            *reinterpret_cast<f32x4*>(&As[buf][(ar + 64 * u) * kLAS + ac]) = ra[u];
            *reinterpret_cast<f32x4*>(&Bs[buf][(br + 8 * u) * kLBS + bc]) = rb[u];
        }
    };

    f32x4 acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

    fetch(0);
    publish(0);
    __syncthreads();
    for (int k0 = 0, buf = 0; k0 < k; k0 += kBK, buf ^= 1) {
        fetch(k0 + kBK);                       // (past the end: zero-filled, published, never read)
        f32x4 af[4];
#pragma unroll
        for (int a = 0; a < 4; ++a) af[a] = *reinterpret_cast<const f32x4*>(&As[buf][(wm + 16 * a + j) * kLAS + 4 * g]);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            float bf[4];
#pragma unroll
            for (int b = 0; b < 4; ++b) bf[b] = Bs[buf][(4 * g + ks) * kLBS + wn + 16 * b + j];
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[a][ks], bf[b], acc[a][b], 0, 0, 0);
        }
        publish(buf ^ 1);                      // last read one barrier ago
        __syncthreads();
    }

#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
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
    // c is written while a and b are still being read (and, with k split over two workgroups,
    // cleared before anything is read): an output that shares storage with an input is refused
    {
        const uintptr_t c0 = reinterpret_cast<uintptr_t>(c), c1 = c0 + (size_t)m * (size_t)n * sizeof(float);
        const uintptr_t a0 = reinterpret_cast<uintptr_t>(a), a1 = a0 + (size_t)m * (size_t)k * sizeof(float);
        const uintptr_t b0 = reinterpret_cast<uintptr_t>(b), b1 = b0 + (size_t)k * (size_t)n * sizeof(float);
        if (k > 0 && ((c0 < a1 && a0 < c1) || (c0 < b1 && b0 < c1)))
            return fail(MYFA_E_INVALID, "matmul: the output overlaps an input (c may not alias a or b)");
    }
    // (the vector path's 32-bit buffer offsets cover 128 rows of A and 16 rows of B)
    const bool vec = k % 4 == 0 && n % 4 == 0 && k <= (1 << 21) && n <= (1 << 24) &&
                     ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b)) % 16 == 0);
    // 128 x 128 blocks (142 TFLOP/s at best against the 64 x 64 kernel's 137) once they fill the chip
    // twice over -- 512 of them run at a time -- and only while their rounds of 512 are at least 90 %
    // full: 3072^3 is 576 blocks, 104 TFLOP/s against 131 with the smaller blocks; 4608^3 118 against
    // 131.  The tuning flavour of the library can force either kernel
    const int force = tuning().matmul_block;
    const int64_t w128 = cdiv(n, kLN) * cdiv(m, kLM);
    const bool large = force ? force == 128 : (w128 >= 512 && w128 * 10 >= cdiv(w128, (int64_t)512) * 512 * 9);
    if (large) {
        const dim3 g128((unsigned)cdiv(n, kLN), (unsigned)cdiv(m, kLM));
        if (vec) launch_kernel(sgemm_mfma_128_kernel<true>, g128, dim3(256), 0, s, a, b, c, (int)m, (int)k, (int)n);
        else     launch_kernel(sgemm_mfma_128_kernel<false>, g128, dim3(256), 0, s, a, b, c, (int)m, (int)k, (int)n);
        return check_launch("matmul_f32");
    }
    // Up to three workgroups per CU and a long k: two workgroups share each block's k range (C zeroed
    // first, two atomic adds per element -- the same value whichever comes first).  Measured:
    // 1024 x 4096 x 1024 115 -> 91 us, 128 x 8192 x 128 201 -> 106 us, 1536^3 85 -> 74 us; 1024^3 unchanged.
    const int64_t w64 = cdiv(n, (int64_t)kBN) * cdiv(m, (int64_t)kBM);
    const int split_knob = tuning().matmul_ksplit;
    const bool split = split_knob ? split_knob > 1 : (w64 <= 768 && k >= 512);
    int kslice = (int)k, zs = 1;
    if (split) {
        kslice = (int)(cdiv(cdiv(k, (int64_t)2), (int64_t)kBK) * kBK);
        zs = (int)cdiv(k, (int64_t)kslice);
        if (zs > 1) {
            if (const hipError_t e = hipMemsetAsync(c, 0, (size_t)m * (size_t)n * sizeof(float), s); e != hipSuccess)
                return fail(MYFA_E_LAUNCH, "matmul: clearing the output: %s", hipGetErrorString(e));
        }
    }
    const dim3 grid((unsigned)cdiv(n, kBN), (unsigned)cdiv(m, kBM), (unsigned)zs);
    if (vec) launch_kernel(sgemm_mfma_kernel<true>, grid, dim3(256), 0, s, a, b, c, (int)m, (int)k, (int)n, kslice);
    else     launch_kernel(sgemm_mfma_kernel<false>, grid, dim3(256), 0, s, a, b, c, (int)m, (int)k, (int)n, kslice);
    return check_launch("matmul_f32");
}

}  // namespace myfa
