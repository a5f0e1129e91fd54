// Issue rates of the vector instructions of the softmax phase, one wave per SIMD (gfx950).
//   hipcc --offload-arch=gfx950 -O3 tools/micro/valu_rates.hip -o /tmp/valu_rates && /tmp/valu_rates
// Each variant runs ITER iterations of an unrolled block on 16 independent registers and reports
// cycles per wave-instruction (s_memtime counts at 100 MHz; the wall clock of the launch is used).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

constexpr int ITER = 4096;

template <int MODE>
__global__ void __launch_bounds__(64) k(float* out, float seed) {
    float r[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) r[i] = seed + 0.001f * (threadIdx.x + i);
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (MODE == 0) asm volatile("v_exp_f32 %0, %0" : "+v"(r[i]));
            if (MODE == 1) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(r[i]));
            if (MODE == 2) { asm volatile("v_exp_f32 %0, %0" : "+v"(r[i])); asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(r[(i + 8) & 15])); }
            if (MODE == 3) { asm volatile("v_exp_f32 %0, %0" : "+v"(r[i])); asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(r[(i + 8) & 15]));
                             asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(r[(i + 4) & 15])); asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(r[(i + 12) & 15])); }
            if (MODE == 4) asm volatile("v_add_f32 %0, %0, %0" : "+v"(r[i]));
            if (MODE == 5) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %0" : "+v"(r[i]));
            if (MODE == 6) asm volatile("v_max3_f32 %0, %0, %0, %0" : "+v"(r[i]));
            if (MODE == 7) asm volatile("v_pk_add_f32 %0, %0, %0" : "+v"(*reinterpret_cast<double*>(&r[(2 * i) & 14])));
            if (MODE == 8) asm volatile("v_dot2c_f32_bf16 %0, %1, %1" : "+v"(r[i]) : "v"(r[(i + 1) & 15]));
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += r[i];
    out[blockIdx.x * 64 + threadIdx.x] = s;
}

template <int MODE>
void run(const char* name, int per_iter, float* out) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    const int grid = 256 * 4;                      // one wave per SIMD
    k<MODE><<<grid, 64>>>(out, 0.5f);
    hipDeviceSynchronize();
    hipEventRecord(a);
    k<MODE><<<grid, 64>>>(out, 0.5f);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0.f;
    hipEventElapsedTime(&ms, a, b);
    int mhz = 0;
    hipDeviceGetAttribute(&mhz, hipDeviceAttributeClockRate, 0);   // kHz
    const double cycles = ms * 1e-3 * (double)mhz * 1e3;
    printf("%-40s %8.3f ms  %6.2f cycles per instruction (at %d MHz nominal)\n", name, ms, cycles / ((double)ITER * per_iter), mhz / 1000);
}

int main() {
    float* out;
    hipMalloc(&out, 256 * 4 * 64 * sizeof(float));
    run<0>("v_exp_f32", 16, out);
    run<1>("v_fma_f32", 16, out);
    run<4>("v_add_f32", 16, out);
    run<5>("v_cvt_pk_bf16_f32", 16, out);
    run<6>("v_max3_f32", 16, out);
    run<7>("v_pk_add_f32", 16, out);
    run<8>("v_dot2c_f32_bf16", 16, out);
    run<2>("v_exp_f32 + v_fma_f32 (pairs)", 32, out);
    run<3>("v_exp_f32 + 3 v_fma_f32", 64, out);
    return 0;
}
