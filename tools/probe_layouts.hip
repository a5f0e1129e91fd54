// Standalone probe of the gfx950 primitives the attention kernel relies on.
// Build: hipcc --offload-arch=gfx950 -O2 tools/probe_layouts.hip -o tools/probe_layouts
// Each check prints PASS/FAIL; exact small-integer data, asymmetric operands.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(2); } } while (0)

// ---- 1. permlane32_swap(x, x): both halves of x in every lane
__global__ void k_swap(float* out) {
    const int lane = threadIdx.x;
    const float x = (float)(lane + 1);
    const unsigned u = __builtin_bit_cast(unsigned, x);
    auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    // copy the elements out first: __builtin_bit_cast(float, r[1]) reads element 0 (hipcc 7.2)
    const unsigned a = r[0], b = r[1];
    out[lane] = __builtin_bit_cast(float, a);
    out[64 + lane] = __builtin_bit_cast(float, b);
}

// ---- 2. MFMA 32x32x16 bf16: C[m][n] = sum_k A[m][k] B[k][n]
// lane (r = l&31, h = l>>5): a[j] = A[r][8h+j], b[j] = B[8h+j][r]; C: col = l&31, row = (i&3)+8(i>>2)+4h
__global__ void k_mfma(const float* A, const float* B, float* C) {
    const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (__bf16)A[r * 16 + 8 * h + j]; b[j] = (__bf16)B[(8 * h + j) * 32 + r]; }
    f32x16 c = {0};
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    for (int i = 0; i < 16; ++i) C[((i & 3) + 8 * (i >> 2) + 4 * h) * 32 + r] = c[i];
}

// ---- 3. accumulator as B operand: X[32x32] from one MFMA, then Y = A2 (32x32) . X, A2 fed in the
// permuted k order: for k-step s, element j of half h <-> row 16s + 8(j>>2) + 4h + (j&3) of X
__global__ void k_chain(const float* A, const float* B, const float* A2, float* Y) {
    const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (__bf16)A[r * 16 + 8 * h + j]; b[j] = (__bf16)B[(8 * h + j) * 32 + r]; }
    f32x16 x = {0};
    x = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, x, 0, 0, 0);
    f32x16 y = {0};
    for (int s = 0; s < 2; ++s) {
        bf16x8 xb, a2;
        for (int j = 0; j < 8; ++j) {
            xb[j] = (__bf16)x[8 * s + j];
            const int krow = 16 * s + 8 * (j >> 2) + 4 * h + (j & 3);
            a2[j] = (__bf16)A2[r * 32 + krow];
        }
        y = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, xb, y, 0, 0, 0);
    }
    for (int i = 0; i < 16; ++i) Y[((i & 3) + 8 * (i >> 2) + 4 * h) * 32 + r] = y[i];
}

// ---- 4. ds_read_b64_tr_b16 on a plain row-major [16 rows][64 cols] tile of shorts, element = row*64+col.
// Group g = lane>>4 reads the 4x16 block at rows 4*(g>>1).., cols 16*(g&1)..; lane 4q+p supplies row q, cols 4p..4p+3.
__global__ void k_tr(short* out) {
    __shared__ __attribute__((aligned(16))) short tile[16 * 64];
    const int lane = threadIdx.x;
    for (int i = lane; i < 16 * 64; i += 64) tile[i] = (short)i;
    __syncthreads();
    const int g = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3;
    const int row = 4 * (g >> 1) + q, col = 16 * (g & 1) + 4 * p;
    s16x4 t = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(tile + row * 64 + col));
    for (int e = 0; e < 4; ++e) out[lane * 4 + e] = t[e];
}

// ---- 5. v_mfma_f64_16x16x4_f64 / v_mfma_f32_16x16x4_f32: D[16x16] = A[16x4] B[4x16] + C
// lane l (j = l&15, g = l>>4): a = A[j][g], b = B[g][j]; D, r = 0..3: f32 holds D[4g + r][j], f64 holds D[4r + g][j]
typedef double f64x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k_mfma64(const float* A, const float* B, float* C64, float* C32) {
    const int lane = threadIdx.x, j = lane & 15, g = lane >> 4;
    f64x4 c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f64_16x16x4f64((double)A[j * 4 + g], (double)B[g * 16 + j], c, 0, 0, 0);
    f32x4 e = {0, 0, 0, 0};
    e = __builtin_amdgcn_mfma_f32_16x16x4f32(A[j * 4 + g], B[g * 16 + j], e, 0, 0, 0);
    for (int r = 0; r < 4; ++r) { C64[(4 * r + g) * 16 + j] = (float)c[r]; C32[(4 * g + r) * 16 + j] = e[r]; }
}
// ---- 6. chain: X[16 keys x 16 queries] from the f64 MFMA; Y[16 x 16] = A2[16 x 16 keys] . X with the
// f32 MFMA, k-step r taking keys 4r .. 4r+3: b = (float)x[r], a = A2[j][4r + g]
__global__ void k_chain64(const float* A, const float* B, const float* A2, float* Y) {
    const int lane = threadIdx.x, j = lane & 15, g = lane >> 4;
    f64x4 x = {0, 0, 0, 0};
    x = __builtin_amdgcn_mfma_f64_16x16x4f64((double)A[j * 4 + g], (double)B[g * 16 + j], x, 0, 0, 0);
    f32x4 y = {0, 0, 0, 0};
    for (int r = 0; r < 4; ++r)
        y = __builtin_amdgcn_mfma_f32_16x16x4f32(A2[j * 16 + 4 * r + g], (float)x[r], y, 0, 0, 0);
    for (int r = 0; r < 4; ++r) Y[(4 * g + r) * 16 + j] = y[r];
}

int main() {
    int fails = 0;
    {   // 1
        float* d; CK(hipMalloc(&d, 128 * 4));
        hipLaunchKernelGGL(k_swap, 1, 64, 0, 0, d); CK(hipDeviceSynchronize());
        float h[128]; CK(hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost));
        printf("swap(u,u): r0[lane0]=%g r0[lane32]=%g r1[lane0]=%g r1[lane32]=%g  (x = lane+1)\n", h[0], h[32], h[64], h[96]);
        bool ok = true;
        for (int l = 0; l < 64; ++l) {
            const float lo = (float)((l & 31) + 1), hi = (float)((l & 31) + 33);
            if (h[l] != lo || h[64 + l] != hi) ok = false;
        }
        printf("1. permlane32_swap(u,u) -> (x_lo, x_hi) in every lane: %s\n", ok ? "PASS" : "FAIL"); fails += !ok;
    }
    std::vector<float> A(32 * 16), B(16 * 32), A2(32 * 32), C(32 * 32), X(32 * 32), Y(32 * 32);
    for (int m = 0; m < 32; ++m) for (int k = 0; k < 16; ++k) A[m * 16 + k] = (float)((m + 2 * k) % 7 - 3);
    for (int k = 0; k < 16; ++k) for (int n = 0; n < 32; ++n) B[k * 32 + n] = (float)((3 * k + n) % 5 - 2);
    for (int m = 0; m < 32; ++m) for (int k = 0; k < 32; ++k) A2[m * 32 + k] = (float)((2 * m + 3 * k) % 4 - 1);
    for (int m = 0; m < 32; ++m) for (int n = 0; n < 32; ++n) { float s = 0; for (int k = 0; k < 16; ++k) s += A[m * 16 + k] * B[k * 32 + n]; X[m * 32 + n] = s; }
    for (int m = 0; m < 32; ++m) for (int n = 0; n < 32; ++n) { float s = 0; for (int k = 0; k < 32; ++k) s += A2[m * 32 + k] * X[k * 32 + n]; Y[m * 32 + n] = s; }
    float *dA, *dB, *dA2, *dC;
    CK(hipMalloc(&dA, A.size() * 4)); CK(hipMalloc(&dB, B.size() * 4)); CK(hipMalloc(&dA2, A2.size() * 4)); CK(hipMalloc(&dC, C.size() * 4));
    CK(hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dA2, A2.data(), A2.size() * 4, hipMemcpyHostToDevice));
    {   // 2
        hipLaunchKernelGGL(k_mfma, 1, 64, 0, 0, dA, dB, dC); CK(hipDeviceSynchronize());
        CK(hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost));
        int bad = 0; for (int i = 0; i < 1024; ++i) bad += C[i] != X[i];
        printf("2. mfma_f32_32x32x16_bf16 A/B/C lane maps: %s (%d mismatches)\n", bad ? "FAIL" : "PASS", bad); fails += bad != 0;
    }
    {   // 3
        hipLaunchKernelGGL(k_chain, 1, 64, 0, 0, dA, dB, dA2, dC); CK(hipDeviceSynchronize());
        CK(hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost));
        int bad = 0; for (int i = 0; i < 1024; ++i) bad += C[i] != Y[i];
        printf("3. accumulator as B operand with permuted k order: %s (%d mismatches)\n", bad ? "FAIL" : "PASS", bad); fails += bad != 0;
    }
    {   // 4
        short* d; CK(hipMalloc(&d, 256 * 2));
        hipLaunchKernelGGL(k_tr, 1, 64, 0, 0, d); CK(hipDeviceSynchronize());
        short h[256]; CK(hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost));
        int bad = 0;
        for (int l = 0; l < 64; ++l) {
            const int g = l >> 4, i = l & 15;
            for (int e = 0; e < 4; ++e) {
                const int want = (4 * (g >> 1) + e) * 64 + 16 * (g & 1) + i;   // row e of the block, column i
                if (h[l * 4 + e] != want) { if (bad < 8) printf("   lane %d elem %d: got %d want %d\n", l, e, h[l * 4 + e], want); ++bad; }
            }
        }
        printf("4. ds_read_b64_tr_b16 (lane i <- column i, element e <- row e): %s (%d mismatches)\n", bad ? "FAIL" : "PASS", bad); fails += bad != 0;
    }
    {   // 5, 6
        std::vector<float> a(16 * 4), b(4 * 16), a2(16 * 16), x(256), y(256), got(256), got2(256);
        for (int m = 0; m < 16; ++m) for (int k = 0; k < 4; ++k) a[m * 4 + k] = (float)((m + 2 * k) % 7 - 3);
        for (int k = 0; k < 4; ++k) for (int n = 0; n < 16; ++n) b[k * 16 + n] = (float)((3 * k + n) % 5 - 2);
        for (int m = 0; m < 16; ++m) for (int k = 0; k < 16; ++k) a2[m * 16 + k] = (float)((2 * m + 3 * k) % 4 - 1);
        for (int m = 0; m < 16; ++m) for (int n = 0; n < 16; ++n) { float t = 0; for (int k = 0; k < 4; ++k) t += a[m * 4 + k] * b[k * 16 + n]; x[m * 16 + n] = t; }
        for (int m = 0; m < 16; ++m) for (int n = 0; n < 16; ++n) { float t = 0; for (int k = 0; k < 16; ++k) t += a2[m * 16 + k] * x[k * 16 + n]; y[m * 16 + n] = t; }
        float *da, *db, *da2, *d1, *d2;
        CK(hipMalloc(&da, 256)); CK(hipMalloc(&db, 256)); CK(hipMalloc(&da2, 1024)); CK(hipMalloc(&d1, 1024)); CK(hipMalloc(&d2, 1024));
        CK(hipMemcpy(da, a.data(), 256, hipMemcpyHostToDevice)); CK(hipMemcpy(db, b.data(), 256, hipMemcpyHostToDevice));
        CK(hipMemcpy(da2, a2.data(), 1024, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_mfma64, 1, 64, 0, 0, da, db, d1, d2); CK(hipDeviceSynchronize());
        CK(hipMemcpy(got.data(), d1, 1024, hipMemcpyDeviceToHost)); CK(hipMemcpy(got2.data(), d2, 1024, hipMemcpyDeviceToHost));
        int bad = 0, bad2 = 0; for (int i = 0; i < 256; ++i) { bad += got[i] != x[i]; bad2 += got2[i] != x[i]; }
        printf("5. mfma_f64_16x16x4 / mfma_f32_16x16x4 lane maps: %s (%d / %d mismatches)\n", (bad || bad2) ? "FAIL" : "PASS", bad, bad2); fails += (bad || bad2) != 0;
        hipLaunchKernelGGL(k_chain64, 1, 64, 0, 0, da, db, da2, d1); CK(hipDeviceSynchronize());
        CK(hipMemcpy(got.data(), d1, 1024, hipMemcpyDeviceToHost));
        bad = 0; for (int i = 0; i < 256; ++i) bad += got[i] != y[i];
        printf("6. f64 accumulator as B operand of the f32 MFMA (keys 4r..4r+3 per step): %s (%d mismatches)\n", bad ? "FAIL" : "PASS", bad); fails += bad != 0;
    }
    printf("%s\n", fails ? "PROBE: FAILURES" : "PROBE: ALL PASS");
    return fails ? 1 : 0;
}
