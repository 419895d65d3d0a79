// Stand-alone timing / ablation harness for csrc/mlp_stream.hpp (tuning tool; not part of the library).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-honor-nans profiles/probes/mlps_probe.hip -o profiles/probes/mlps_probe
//   ./mlps_probe [M]      -> time of the streamed C = 128 MLP forward and of its ablations (ABL bits, see mlp_stream.hpp)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <vector>
#include "../../rvt_amd/csrc/mlp_stream.hpp"
namespace rvt { void set_last_error(const char*, ...) {} int check_launch(const char*) { return 0; } ::RvtTuning g_tuning = RVT_TUNING_DEFAULTS; }
using namespace rvt;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__global__ void fill_kernel(bf16* p, size_t n, unsigned seed, float scale) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        unsigned h = (unsigned)i * 2654435761u + seed; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        p[i] = (bf16)(((float)(h & 0xffff) / 32768.0f - 1.0f) * scale);
    }
}
__global__ void fillf_kernel(float* p, size_t n, float v) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) p[i] = v;
}
struct Bufs { bf16 *x, *y, *dy, *w1, *w2, *w2gt; float *lw, *lb, *b1, *b2, *gam, *dlw, *dlb; int M; };

template <int ABL> static float run_fwd(const Bufs& b, int iters) {
    auto k = mlps_fwd_kernel<bf16, 128, 8, 2, ABL>;
    const int grid = 256;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 2; i++) hipLaunchKernelGGL(k, dim3(grid), dim3(512), 0, 0, b.x, b.y, b.lw, b.lb, b.w1, b.b1, b.w2, b.b2, b.gam, b.M, 1e-5f);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < iters; i++) hipLaunchKernelGGL(k, dim3(grid), dim3(512), 0, 0, b.x, b.y, b.lw, b.lb, b.w1, b.b1, b.w2, b.b2, b.gam, b.M, 1e-5f);
    CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / iters;
}

int main(int argc, char** argv) {
    Bufs b; b.M = argc > 1 ? atoi(argv[1]) : 1935360;
    const int C = 128, M = b.M;
    CK(hipMalloc(&b.x, (size_t)M * C * 2)); CK(hipMalloc(&b.y, (size_t)M * C * 2)); CK(hipMalloc(&b.dy, (size_t)M * C * 2));
    CK(hipMalloc(&b.w1, 4 * C * C * 2)); CK(hipMalloc(&b.w2, 4 * C * C * 2)); CK(hipMalloc(&b.w2gt, 4 * C * C * 2));
    float* f; CK(hipMalloc(&f, 16 * C * 4));
    b.lw = f; b.lb = f + C; b.b2 = f + 2 * C; b.gam = f + 3 * C; b.b1 = f + 4 * C; b.dlw = f + 8 * C; b.dlb = f + 9 * C;
    hipLaunchKernelGGL(fill_kernel, dim3(1024), dim3(256), 0, 0, b.x, (size_t)M * C, 1u, 1.5f);
    hipLaunchKernelGGL(fill_kernel, dim3(1024), dim3(256), 0, 0, b.dy, (size_t)M * C, 3u, 1.0f);
    hipLaunchKernelGGL(fill_kernel, dim3(64), dim3(256), 0, 0, b.w1, (size_t)4 * C * C, 7u, 0.1f);
    hipLaunchKernelGGL(fill_kernel, dim3(64), dim3(256), 0, 0, b.w2, (size_t)4 * C * C, 9u, 0.1f);
    hipLaunchKernelGGL(fill_kernel, dim3(64), dim3(256), 0, 0, b.w2gt, (size_t)4 * C * C, 11u, 0.1f);
    hipLaunchKernelGGL(fillf_kernel, dim3(8), dim3(256), 0, 0, f, (size_t)16 * C, 0.0f);
    hipLaunchKernelGGL(fillf_kernel, dim3(1), dim3(256), 0, 0, b.lw, (size_t)C, 1.0f);
    hipLaunchKernelGGL(fillf_kernel, dim3(1), dim3(256), 0, 0, b.gam, (size_t)C, 1.0f);
    CK(hipDeviceSynchronize());
    const double fl = 16.0 * M * C * C;
    struct { const char* name; float ms; } r[] = {
        {"forward, full kernel", run_fwd<0>(b, 10)},
        {"no weight stream (1)", run_fwd<1>(b, 10)},
        {"no barrier (2)", run_fwd<2>(b, 10)},
        {"no stream, no barrier (3)", run_fwd<3>(b, 10)},
        {"no table gather (4)", run_fwd<4>(b, 10)},
        {"no stream, no barrier, no gather (7)", run_fwd<7>(b, 10)},
        {"forward, full kernel (again)", run_fwd<0>(b, 10)},
        {"v1: 7 + no fragment reads (15)", run_fwd<15>(b, 10)},
        {"v1: no fragment reads only (8)", run_fwd<8>(b, 10)},
        {"v1: 15 + split fc1 chain (31)", run_fwd<31>(b, 10)},
        {"v1: split fc1 chain only (16)", run_fwd<16>(b, 10)},
    };
    printf("mlps forward M = %d (%.1f GFLOP)\n", M, fl * 1e-9);
    for (auto& x : r) printf("  %-40s %8.3f ms   %7.1f TFLOP/s\n", x.name, x.ms, fl / x.ms * 1e-9);
    return 0;
}
