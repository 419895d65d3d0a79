// Does ONE wave overlap its own MFMAs with its own independent VALU work on gfx950?  (round 6)
// Three loops per wave, one wave per SIMD (a 96 KiB LDS block keeps the second workgroup off the CU):
//   mfma : 8 independent v_mfma_f32_32x32x16_bf16 per iteration
//   valu : 64 dependent-free v_exp_f32 / v_rcp_f32 / fma per iteration (the gate math mix of the ConvLSTM scans)
//   both : the two interleaved in source order (MFMA, 8 VALU, MFMA, 8 VALU ...) - the compiler keeps that order (sched_barrier)
// If both ~ max(mfma, valu) the issue slot is shared fine; if both ~ mfma + valu an MFMA holds its wave's issue slot.
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_valu_overlap_probe mfma_valu_overlap_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int MODE>
__global__ void __launch_bounds__(256) probe(float* out, int iters) {
    __shared__ char hog[96 * 1024];
    if (threadIdx.x == 0 && iters < 0) hog[0] = 1;
    f32x16 acc[8];
    bf16x8 a, b;
    for (int i = 0; i < 8; i++) { a[i] = (__bf16)(0.001f * (threadIdx.x + i)); b[i] = (__bf16)(0.002f * i); }
    for (int j = 0; j < 8; j++) for (int i = 0; i < 16; i++) acc[j][i] = 0.f;
    float v[16];
    for (int i = 0; i < 16; i++) v[i] = 0.01f * (threadIdx.x + i);
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int j = 0; j < 8; j++) {
            if (MODE & 1) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[j], 0, 0, 0);
            if (MODE & 2) {
#pragma unroll
                for (int q = 0; q < 2; q++) {      // 2 x (exp, add, rcp, fma) = 8 VALU, 4 of them transcendental
                    float& x = v[2 * j + q];
                    x = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * -1.44f)) * 1.0001f + 0.0001f;
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = 0.f;
    for (int j = 0; j < 8; j++) for (int i = 0; i < 16; i++) s += acc[j][i];
    for (int i = 0; i < 16; i++) s += v[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE> float run(float* out, int iters) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(probe<MODE>, dim3(256), dim3(256), 0, 0, out, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(probe<MODE>, dim3(256), dim3(256), 0, 0, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}
int main() {
    float* out; hipMalloc(&out, 256 * 256 * 4);
    const int iters = 20000;
    const float m = run<1>(out, iters), v = run<2>(out, iters), b = run<3>(out, iters);
    const double clk = 2.4e9;
    printf("per iteration (8 MFMA | 64 VALU incl. 32 transcendental), one wave per SIMD:\n");
    printf("  mfma only %.3f ms = %.0f cycles/iter\n  valu only %.3f ms = %.0f cycles/iter\n  both      %.3f ms = %.0f cycles/iter   (sum %.0f, max %.0f)\n",
           m, m * 1e-3 * clk / iters, v, v * 1e-3 * clk / iters, b, b * 1e-3 * clk / iters, (m + v) * 1e-3 * clk / iters, (m > v ? m : v) * 1e-3 * clk / iters);
    return 0;
}
