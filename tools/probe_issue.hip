// Issue-rate probe for gfx950: how MFMA and VALU streams of two waves on one SIMD share time.
// Build: hipcc --offload-arch=gfx950 -O2 tools/probe_issue.hip -o tools/probe_issue
// One workgroup of 512 threads = 2 waves per SIMD.  Waves 0-3 run stream A, waves 4-7 stream B,
// both started together behind a barrier; each prints shader cycles per instruction.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(2); } } while (0)

__device__ __forceinline__ unsigned long long stamp() {
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
    return t;
}

enum { NONE = 0, MFMA32, MFMA16, FMA, EXP, PKFMA, PKADD, CVT, MAX3, SOFTMAX, SOFTMAX_PK, LDSR, FMA_DEP, EXP_DEP, MAX3_DEP, SOFTMAX_SKEW, SOFTMAX_BLOCK, MFMA_F64, MFMA_F32, CVT_F64, NMODES };
static const char* kNames[NMODES] = {"none", "mfma32x32x16", "mfma16x16x32", "v_fma", "v_exp", "v_pk_fma", "v_pk_add",
                                     "v_cvt_pk_bf16", "v_max3", "softmax(fma,exp,add)", "softmax(pk)", "ds_read_b128", "v_fma dependent", "v_exp dependent", "v_max3 dependent",
                                     "softmax skewed", "softmax blocked", "mfma_f64_16x16x4", "mfma_f32_16x16x4", "v_cvt_f64_f32"};
// instructions per outer iteration for each mode
__host__ __device__ constexpr int per_iter(int mode) {
    return mode == MFMA32 || mode == MFMA_F64 || mode == MFMA_F32 ? 16 : mode == MFMA16 ? 16 : mode == SOFTMAX || mode == SOFTMAX_SKEW || mode == SOFTMAX_BLOCK ? 384 : mode == SOFTMAX_PK ? 256 : mode == NONE ? 0 : mode == LDSR ? 32 : 128;
}

template <int MODE>
__device__ __forceinline__ float run_stream(int iters, float seed, char* lds, unsigned long long& half) {
    float acc = 0.f;
    if constexpr (MODE == MFMA32) {
        bf16x8 a, b;
        for (int j = 0; j < 8; ++j) { a[j] = (__bf16)seed; b[j] = (__bf16)(seed + 1.f); }
        f32x16 c[4] = {};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 16; ++u) c[u & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c[u & 3], 0, 0, 0);
        }
        for (int u = 0; u < 4; ++u) acc += c[u][0];
    } else if constexpr (MODE == MFMA16) {
        bf16x8 a, b;
        for (int j = 0; j < 8; ++j) { a[j] = (__bf16)seed; b[j] = (__bf16)(seed + 1.f); }
        f32x4 c[4] = {};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 16; ++u) c[u & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c[u & 3], 0, 0, 0);
        }
        for (int u = 0; u < 4; ++u) acc += c[u][0];
    } else if constexpr (MODE == MFMA_F64) {
        typedef double f64x4 __attribute__((ext_vector_type(4)));
        f64x4 c[4] = {};
        const double a = seed, b = seed + 1.0;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 16; ++u) c[u & 3] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c[u & 3], 0, 0, 0);
        }
        for (int u = 0; u < 4; ++u) acc += (float)c[u][0];
    } else if constexpr (MODE == MFMA_F32) {
        f32x4 c[4] = {};
        const float a = seed, b = seed + 1.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 16; ++u) c[u & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c[u & 3], 0, 0, 0);
        }
        for (int u = 0; u < 4; ++u) acc += c[u][0];
    } else if constexpr (MODE == LDSR) {
        f32x4 x[4] = {};
        const int off = (threadIdx.x & 63) * 16;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 32; ++u) {
                f32x4 v;
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(off), "n"(0));
                asm volatile("s_waitcnt lgkmcnt(4)");
                x[u & 3] = v;
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)");
        acc = x[0][0] + x[1][0] + x[2][0] + x[3][0];
    } else {
        float x[8];
        f32x2 p[4];
        for (int u = 0; u < 8; ++u) x[u] = seed * 0.001f * (u + 1);
        for (int u = 0; u < 4; ++u) p[u] = f32x2{x[u], x[u + 4]};
        const float sc = 0.999f, mm = 0.0001f;
        const f32x2 sc2 = {sc, sc}, mm2 = {mm, mm};
        unsigned pk[4] = {};
        for (int it = 0; it < iters; ++it) {
            if (it == (iters >> 1)) half = stamp();
#pragma unroll
            for (int u = 0; u < 128; ++u) {
                const int k = u & 7;
                if constexpr (MODE == FMA) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[k]) : "v"(sc), "v"(mm));
                if constexpr (MODE == EXP) asm volatile("v_exp_f32 %0, %0" : "+v"(x[k]));
                if constexpr (MODE == PKFMA) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[k & 3]) : "v"(sc2), "v"(mm2));
                if constexpr (MODE == PKADD) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[k & 3]) : "v"(mm2));
                if constexpr (MODE == CVT) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(pk[k & 3]) : "v"(x[k]), "v"(x[(k + 1) & 7]));
                if constexpr (MODE == MAX3) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(x[k]) : "v"(x[(k + 1) & 7]), "v"(x[(k + 2) & 7]));
                if constexpr (MODE == SOFTMAX) {
                    asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[k]) : "v"(sc), "v"(mm));
                    asm volatile("v_exp_f32 %0, %0" : "+v"(x[k]));
                    asm volatile("v_add_f32 %0, %0, %1" : "+v"(acc) : "v"(x[k]));
                }
                if constexpr (MODE == CVT_F64) { double dd; asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(dd) : "v"(x[k])); if (u == 127) acc += (float)dd; }
                if constexpr (MODE == FMA_DEP) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[0]) : "v"(sc), "v"(mm));
                if constexpr (MODE == EXP_DEP) asm volatile("v_exp_f32 %0, %0" : "+v"(x[0]));
                if constexpr (MODE == MAX3_DEP) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(x[0]) : "v"(x[(k + 1) & 7]), "v"(x[(k + 2) & 7]));
                if constexpr (MODE == SOFTMAX_SKEW) {   // fma(k+2), exp(k+1), add(k): nothing reads the previous result
                    asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[(k + 2) & 7]) : "v"(sc), "v"(mm));
                    asm volatile("v_exp_f32 %0, %0" : "+v"(x[(k + 1) & 7]));
                    asm volatile("v_add_f32 %0, %0, %1" : "+v"(p[k & 3].x) : "v"(x[k]));
                }
                if constexpr (MODE == SOFTMAX_BLOCK) {  // 8 fma, then 8 exp, then 8 add on 4 accumulators
                    if ((u & 7) == 0) {
#pragma unroll
                        for (int j = 0; j < 8; ++j) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[j]) : "v"(sc), "v"(mm));
#pragma unroll
                        for (int j = 0; j < 8; ++j) asm volatile("v_exp_f32 %0, %0" : "+v"(x[j]));
#pragma unroll
                        for (int j = 0; j < 8; ++j) asm volatile("v_add_f32 %0, %0, %1" : "+v"(p[j & 3].x) : "v"(x[j]));
                    }
                }
                if constexpr (MODE == SOFTMAX_PK) {     // per PAIR of elements: pk_fma, 2 exp, pk_add = 4 instr
                    if (u < 64) {
                        asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[k & 3]) : "v"(sc2), "v"(mm2));
                        asm volatile("v_exp_f32 %0, %0" : "+v"(p[k & 3].x));
                        asm volatile("v_exp_f32 %0, %0" : "+v"(p[k & 3].y));
                        asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[(k + 2) & 3]) : "v"(p[k & 3]));
                    }
                }
            }
        }
        for (int u = 0; u < 8; ++u) acc += x[u];
        for (int u = 0; u < 4; ++u) acc += p[u].x + p[u].y + (float)pk[u];
    }
    (void)lds;
    return acc;
}

template <int A, int B, int PA = 0, int PB = 0>
__global__ void __launch_bounds__(512) k_issue(int iters_a, int iters_b, unsigned long long* cyc, float* sink) {
    __shared__ __attribute__((aligned(16))) char lds[4096];
    for (int i = threadIdx.x; i < 1024; i += 512) reinterpret_cast<float*>(lds)[i] = (float)i;
    __syncthreads();
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const float seed = (float)(threadIdx.x & 3) + 1.f;
    float r;
    unsigned long long t0, t1, th = 0;
    if (wave < 4) {
        __builtin_amdgcn_s_setprio(PA);
        t0 = stamp();
        r = run_stream<A>(iters_a, seed, lds, th);
        asm volatile("s_nop 0" :: "v"(r));
        t1 = stamp();
    } else {
        __builtin_amdgcn_s_setprio(PB);
        t0 = stamp();
        r = run_stream<B>(iters_b, seed, lds, th);
        asm volatile("s_nop 0" :: "v"(r));
        t1 = stamp();
    }
    if ((threadIdx.x & 63) == 0) { cyc[wave] = t1 - t0; cyc[8 + wave] = th ? th - t0 : 0; }
    if (r == 123.456f) sink[threadIdx.x] = r;
}

template <int A, int B, int PA = 0, int PB = 0>
void run(int ia, int ib, unsigned long long* d_cyc, float* d_sink) {
    CK(hipMemset(d_cyc, 0, 16 * sizeof(unsigned long long)));
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((k_issue<A, B, PA, PB>), dim3(1), dim3(512), 0, 0, ia, ib, d_cyc, d_sink);
    CK(hipDeviceSynchronize());
    unsigned long long h[16];
    CK(hipMemcpy(h, d_cyc, sizeof(h), hipMemcpyDeviceToHost));
    const double ca = per_iter(A) ? (double)h[0] / ((double)ia * per_iter(A)) : 0.0;
    const double cb = per_iter(B) ? (double)h[4] / ((double)ib * per_iter(B)) : 0.0;
    printf("prio %d/%d A=%-22s B=%-22s  A: %7.2f cyc/instr (total %8llu)   B: %7.2f cyc/instr (total %8llu, first half %8llu)\n",
           PA, PB, kNames[A], kNames[B], ca, h[0], cb, h[4], h[12]);
}

int main() {
    unsigned long long* d_cyc; float* d_sink;
    CK(hipMalloc(&d_cyc, 128)); CK(hipMalloc(&d_sink, 4096));
    const int it = 2000;
    // each stream alone
    run<MFMA32, NONE>(it, 0, d_cyc, d_sink);
    run<MFMA16, NONE>(it, 0, d_cyc, d_sink);
    run<NONE, FMA>(0, it, d_cyc, d_sink);
    run<NONE, EXP>(0, it, d_cyc, d_sink);
    run<NONE, PKFMA>(0, it, d_cyc, d_sink);
    run<NONE, PKADD>(0, it, d_cyc, d_sink);
    run<NONE, CVT>(0, it, d_cyc, d_sink);
    run<NONE, MAX3>(0, it, d_cyc, d_sink);
    run<NONE, SOFTMAX>(0, it, d_cyc, d_sink);
    run<NONE, SOFTMAX_PK>(0, it, d_cyc, d_sink);
    run<NONE, LDSR>(0, it, d_cyc, d_sink);
    run<MFMA_F64, NONE>(it, 0, d_cyc, d_sink);
    run<MFMA_F32, NONE>(it, 0, d_cyc, d_sink);
    run<NONE, CVT_F64>(0, it, d_cyc, d_sink);
    run<MFMA_F64, MFMA_F64>(it, it, d_cyc, d_sink);
    run<NONE, FMA_DEP>(0, it, d_cyc, d_sink);
    run<NONE, EXP_DEP>(0, it, d_cyc, d_sink);
    run<NONE, MAX3_DEP>(0, it, d_cyc, d_sink);
    run<NONE, SOFTMAX_SKEW>(0, it, d_cyc, d_sink);
    run<NONE, SOFTMAX_BLOCK>(0, it, d_cyc, d_sink);
    // same stream in both waves of a SIMD
    run<MFMA32, MFMA32>(it, it, d_cyc, d_sink);
    run<MFMA16, MFMA16>(it, it, d_cyc, d_sink);
    run<FMA, FMA>(it, it, d_cyc, d_sink);
    run<EXP, EXP>(it, it, d_cyc, d_sink);
    // MFMA in one wave against VALU in the other (the ping-pong situation).
    // "B inside A": B is short, so its figure is its rate under MFMA load;
    // "A inside B": A is short, so its figure is the MFMA rate under VALU load.
    printf("-- B inside A\n");
    run<MFMA32, FMA>(16 * it, it / 4, d_cyc, d_sink);
    run<MFMA32, EXP>(16 * it, it / 4, d_cyc, d_sink);
    run<MFMA32, PKFMA>(16 * it, it / 4, d_cyc, d_sink);
    run<MFMA32, SOFTMAX>(16 * it, it / 4, d_cyc, d_sink);
    run<MFMA32, SOFTMAX_PK>(16 * it, it / 4, d_cyc, d_sink);
    run<MFMA32, LDSR>(16 * it, it / 4, d_cyc, d_sink);
    run<MFMA16, FMA>(16 * it, it / 4, d_cyc, d_sink);
    run<MFMA16, EXP>(16 * it, it / 4, d_cyc, d_sink);
    run<MFMA16, SOFTMAX>(16 * it, it / 4, d_cyc, d_sink);
    printf("-- B inside A, B at higher priority\n");
    run<MFMA32, FMA, 0, 3>(16 * it, it / 4, d_cyc, d_sink);
    run<MFMA32, EXP, 0, 3>(16 * it, it / 4, d_cyc, d_sink);
    run<MFMA32, SOFTMAX, 0, 3>(16 * it, it / 4, d_cyc, d_sink);
    run<MFMA32, SOFTMAX_SKEW, 0, 3>(16 * it, it / 4, d_cyc, d_sink);
    run<MFMA32, SOFTMAX_BLOCK, 0, 3>(16 * it, it / 4, d_cyc, d_sink);
    run<MFMA32, SOFTMAX_PK, 0, 3>(16 * it, it / 4, d_cyc, d_sink);
    run<MFMA16, SOFTMAX, 0, 3>(16 * it, it / 4, d_cyc, d_sink);
    printf("-- A inside B, B at higher priority\n");
    run<MFMA32, FMA, 0, 3>(it, 8 * it, d_cyc, d_sink);
    run<MFMA32, EXP, 0, 3>(it, 8 * it, d_cyc, d_sink);
    run<MFMA32, SOFTMAX, 0, 3>(it, 4 * it, d_cyc, d_sink);
    run<MFMA16, SOFTMAX, 0, 3>(it, 2 * it, d_cyc, d_sink);
    printf("-- B inside A, A at higher priority\n");
    run<MFMA32, FMA, 3, 0>(16 * it, it / 4, d_cyc, d_sink);
    run<MFMA32, SOFTMAX, 3, 0>(16 * it, it / 4, d_cyc, d_sink);
    return 0;
}
