// Stand-alone timing / ablation harness for csrc/ppgemm_tn.hpp (tuning tool; not part of the library).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <vector>
#include "../../rvt_amd/csrc/ppgemm_tn.hpp"
namespace rvt { void set_last_error(const char*, ...) {} int check_launch(const char*) { return 0; } }
using namespace rvt;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
__global__ void fill_kernel(bf16* p, size_t n, unsigned seed, float scale) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        unsigned h = (unsigned)i * 2654435761u + seed; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        p[i] = (bf16)(((float)(h & 0xffff) / 32768.0f - 1.0f) * scale);
    }
}
template <int ABL> static float run(const bf16* dY, const bf16* X, float* ws, int M, int N, int K, int iters) {
    const int ns = ppgemm_tn_slices(M, N, K);
    const int tps = (((M + ns - 1) / ns) + 63) / 64 * 64;
    const int ns_eff = (M + tps - 1) / tps;
    const int n_tiles = N / 256, k_tiles = K / 256;
    float* ws_cs = ws + (size_t)ns * N * K;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto go = [&]() { hipLaunchKernelGGL((ppgemm_tn_kernel<ABL>), dim3(8 * n_tiles * k_tiles * ((ns_eff + 7) / 8)), dim3(512), 0, 0, dY, N, X, X, K, K, ws, ws_cs, M, N, K, n_tiles, k_tiles, tps); };
    for (int i = 0; i < 3; i++) go();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < iters; i++) go();
    CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / iters;
}
int main(int argc, char** argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 120960, N = argc > 2 ? atoi(argv[2]) : 2048, K = argc > 3 ? atoi(argv[3]) : 512;
    bf16 *dY, *X; float* ws;
    CK(hipMalloc(&dY, (size_t)M * N * 2)); CK(hipMalloc(&X, (size_t)M * K * 2));
    CK(hipMalloc(&ws, ppgemm_tn_ws_floats(M, N, K, 1) * 4));
    hipLaunchKernelGGL(fill_kernel, dim3(1024), dim3(256), 0, 0, dY, (size_t)M * N, 1u, 0.5f);
    hipLaunchKernelGGL(fill_kernel, dim3(1024), dim3(256), 0, 0, X, (size_t)M * K, 7u, 1.0f);
    CK(hipDeviceSynchronize());
    const double fl = 2.0 * M * N * K;
    struct { const char* name; float ms; } r[] = {
        {"full kernel (no reduce)", run<0>(dY, X, ws, M, N, K, 20)},
        {"no epilogue (16)", run<16>(dY, X, ws, M, N, K, 20)},
        {"no LDS-DMA (1)", run<1>(dY, X, ws, M, N, K, 20)},
        {"no LDS-DMA, no epilogue (1+16)", run<17>(dY, X, ws, M, N, K, 20)},
        {"no barriers (8)", run<8>(dY, X, ws, M, N, K, 20)},
        {"no fragment reads (2+16)", run<18>(dY, X, ws, M, N, K, 20)},
        {"no MFMAs (4+16)", run<20>(dY, X, ws, M, N, K, 20)},
        {"LDS-DMA + barriers only (2+4+16)", run<22>(dY, X, ws, M, N, K, 20)},
    };
    printf("ppgemm_tn %d tokens, dW %d x %d (%.1f GFLOP), %d slices\n", M, N, K, fl * 1e-9, ppgemm_tn_slices(M, N, K));
    for (auto& x : r) printf("  %-36s %8.3f ms   %7.1f TFLOP/s\n", x.name, x.ms, fl / x.ms * 1e-9);
    return 0;
}
