// Empirical map of gfx950's ds_read_b64_tr_b16 (LDS transpose read): which 16-bit LDS elements does lane l receive,
// as a function of the per-lane addresses?  Build: hipcc --offload-arch=gfx950 -O2 tr_read_probe.hip -o tr_read_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

__global__ void probe(const int* __restrict__ lane_addr_bytes, uint16_t* __restrict__ out) {
    __shared__ __attribute__((aligned(16))) uint16_t lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (uint16_t)i;     // value = element index
    __syncthreads();
    const unsigned addr = (unsigned)(uintptr_t)lds + (unsigned)lane_addr_bytes[threadIdx.x];
    uint64_t v;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
    for (int j = 0; j < 4; j++) out[threadIdx.x * 4 + j] = (uint16_t)(v >> (16 * j));
}

static void run(const char* name, const std::vector<int>& addr) {
    int* da; uint16_t* dout;
    hipMalloc(&da, 64 * 4); hipMalloc(&dout, 256 * 2);
    hipMemcpy(da, addr.data(), 64 * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, da, dout);
    std::vector<uint16_t> o(256);
    hipMemcpy(o.data(), dout, 512, hipMemcpyDeviceToHost);
    printf("== %s  (lane: byte address -> the 4 element indices it received)\n", name);
    for (int l = 0; l < 64; l++) {
        printf("  l%02d a=%4d -> %4d %4d %4d %4d%s", l, addr[l], o[l * 4], o[l * 4 + 1], o[l * 4 + 2], o[l * 4 + 3], (l % 2) ? "\n" : "   |");
    }
    hipFree(da); hipFree(dout);
}

int main() {
    std::vector<int> a(64);
    for (int l = 0; l < 64; l++) a[l] = l * 8;                       // lane-linear 8-byte chunks
    run("lane-linear: addr = 8*l", a);
    for (int l = 0; l < 64; l++) a[l] = (l % 16) * 128 + (l / 16) * 8;   // 16 rows of 128 B, 4 column groups
    run("rows: addr = 128*(l%16) + 8*(l/16)", a);
    for (int l = 0; l < 64; l++) a[l] = (l % 16) * 8 + (l / 16) * 512;
    run("addr = 8*(l%16) + 512*(l/16)", a);
    return 0;
}
