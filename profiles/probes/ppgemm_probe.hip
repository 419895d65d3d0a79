// Stand-alone timing / ablation harness for csrc/ppgemm.hpp (tuning tool; not part of the library).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-honor-nans profiles/probes/ppgemm_probe.hip -o profiles/probes/ppgemm_probe
//   ./ppgemm_probe M N K        -> time of the full kernel and of its ablations (ABL bits, see ppgemm.hpp), TFLOP/s, and a spot check
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <vector>
#include "../../rvt_amd/csrc/ppgemm.hpp"
namespace rvt { void set_last_error(const char*, ...) {} int check_launch(const char*) { return 0; } }
using namespace rvt;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__global__ void fill_kernel(bf16* p, size_t n, unsigned seed, float scale) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        unsigned h = (unsigned)i * 2654435761u + seed; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        p[i] = (bf16)(((float)(h & 0xffff) / 32768.0f - 1.0f) * scale);
    }
}
__global__ void ref_kernel(const bf16* X, const bf16* W, const int* rows, int nrows, int N, int K, float* out) {
    const int r = blockIdx.x, m = rows[r];
    for (int n = threadIdx.x; n < N; n += 256) {
        float s = 0.f;
        for (int k = 0; k < K; k++) s += (float)X[(size_t)m * K + k] * (float)W[(size_t)n * K + k];
        out[(size_t)r * N + n] = s;
    }
}

template <int ABL> static float run(const PPMat& X, const PPMat& W, bf16* Y, int M, int N, int K, int iters) {
    PPEpArgs ep{Y, nullptr, nullptr, nullptr, nullptr, N};
    const int m_tiles = (M + 255) / 256, n_tiles = N / 256;
    int grid = 256; const int total = m_tiles * n_tiles;
    { const int per_xcd = ((m_tiles + 7) / 8) * n_tiles; if (grid > 8 * per_xcd) grid = 8 * per_xcd; }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; i++) hipLaunchKernelGGL((ppgemm_kernel<PP_STORE, ABL>), dim3(grid), dim3(512), 0, 0, X, W, ep, M, N, K, m_tiles, n_tiles, PPConv());
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < iters; i++) hipLaunchKernelGGL((ppgemm_kernel<PP_STORE, ABL>), dim3(grid), dim3(512), 0, 0, X, W, ep, M, N, K, m_tiles, n_tiles, PPConv());
    CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / iters;
}

int main(int argc, char** argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 120960, N = argc > 2 ? atoi(argv[2]) : 2048, K = argc > 3 ? atoi(argv[3]) : 512;
    bf16 *X, *W, *Y;
    CK(hipMalloc(&X, (size_t)M * K * 2)); CK(hipMalloc(&W, (size_t)N * K * 2)); CK(hipMalloc(&Y, (size_t)M * N * 2));
    hipLaunchKernelGGL(fill_kernel, dim3(1024), dim3(256), 0, 0, X, (size_t)M * K, 1u, 1.0f);
    hipLaunchKernelGGL(fill_kernel, dim3(1024), dim3(256), 0, 0, W, (size_t)N * K, 7u, 0.05f);
    CK(hipDeviceSynchronize());
    const PPMat xs{X, X, K, K}, ws{W, W, K, K};
    const double fl = 2.0 * M * N * K;
    struct { const char* name; float ms; } r[] = {
        {"full kernel", run<0>(xs, ws, Y, M, N, K, 20)},
        {"no epilogue (16)", run<16>(xs, ws, Y, M, N, K, 20)},
        {"epilogue, stores dropped (32)", run<32>(xs, ws, Y, M, N, K, 20)},
        {"lax wait counts after stores (64)", run<64>(xs, ws, Y, M, N, K, 20)},
        {"no LDS-DMA, stores dropped (1+32)", run<33>(xs, ws, Y, M, N, K, 20)},
        {"no LDS-DMA (1)", run<1>(xs, ws, Y, M, N, K, 20)},
        {"MFMAs + barriers only (1+2+16)", run<19>(xs, ws, Y, M, N, K, 20)},
        {"MFMAs only (1+2+8+16)", run<27>(xs, ws, Y, M, N, K, 20)},
        {"LDS-DMA + barriers only (2+4+16)", run<22>(xs, ws, Y, M, N, K, 20)},
        {"reads + MFMAs + barriers (1+16)", run<17>(xs, ws, Y, M, N, K, 20)},
    };
    printf("ppgemm %d x %d x %d  (%.1f GFLOP)\n", M, N, K, fl * 1e-9);
    for (auto& x : r) printf("  %-36s %8.3f ms   %7.1f TFLOP/s\n", x.name, x.ms, fl / x.ms * 1e-9);
    // spot check of the full kernel
    run<0>(xs, ws, Y, M, N, K, 1);
    const int NR = 64; std::vector<int> rows(NR);
    for (int i = 0; i < NR; i++) rows[i] = (int)(((long long)i * 7919 * 131) % M);
    rows[NR - 1] = M - 1;
    int* drows; float* dref; CK(hipMalloc(&drows, NR * 4)); CK(hipMalloc(&dref, (size_t)NR * N * 4));
    CK(hipMemcpy(drows, rows.data(), NR * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(ref_kernel, dim3(NR), dim3(256), 0, 0, X, W, drows, NR, N, K, dref);
    std::vector<float> ref((size_t)NR * N); std::vector<unsigned short> y((size_t)N);
    CK(hipMemcpy(ref.data(), dref, (size_t)NR * N * 4, hipMemcpyDeviceToHost));
    double worst = 0, scale = 0;
    for (int i = 0; i < NR; i++) {
        CK(hipMemcpy(y.data(), Y + (size_t)rows[i] * N, (size_t)N * 2, hipMemcpyDeviceToHost));
        for (int n = 0; n < N; n++) {
            unsigned u = (unsigned)y[n] << 16; float f; memcpy(&f, &u, 4);
            worst = fmax(worst, fabs(f - ref[(size_t)i * N + n])); scale = fmax(scale, fabs(ref[(size_t)i * N + n]));
        }
    }
    printf("  spot check: max |err| / scale = %.2e\n", worst / scale);
    return 0;
}
