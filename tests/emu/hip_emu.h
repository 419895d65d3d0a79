// SIMT emulator for running the HIP kernel SOURCES of rvt_amd/csrc on a CPU.
//
// TEST INFRASTRUCTURE ONLY.  The authoring container has no GPU and GPU time is rationed, so the
// kernel sources are additionally compiled for x86 (amdclang++ -x c++ -DRVT_EMU -include hip_emu.h)
// into tests/emu/librvt_emu.so.  Each workgroup runs as a set of fibers (one per work-item) on ONE OS
// thread; __syncthreads() is a fiber barrier, and wave-level operations (shuffles, MFMA) are a 64-fiber
// rendezvous that applies the documented gfx950 lane layouts.  This checks index arithmetic, tile
// logic and barrier placement of the real kernel code.  It is never loaded by the rvt_amd package:
// the product path loads only the gfx950 code object and raises when it (or a GPU) is missing.
#pragma once
#define RVT_EMU 1
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __shared__ static

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
typedef void* hipStream_t;
typedef int hipError_t;
#define hipSuccess 0
inline hipError_t hipGetLastError() { return 0; }
inline const char* hipGetErrorString(hipError_t) { return "emu"; }
inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) { memset(p, v, n); return 0; }
inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, int, hipStream_t) { memcpy(d, s, n); return 0; }
#define hipMemcpyDeviceToDevice 3

namespace emu {
enum { RUNNABLE = 0, WAIT_WAVE = 1, WAIT_BLOCK = 2, DONE = 3 };
struct Fiber {
    void* sp;
    dim3 tid;
    int lane, wave, state;
    unsigned wait_gen;
    char* stack;
};
struct Group { unsigned count, arrived, gen; };
struct Ctx {
    dim3 bIdx, bDim, gDim;
    Fiber* cur;
    void* sched_sp;
    Fiber* fibers;
    int nfibers;
    Group block;
    Group waves[16];
    std::function<void()> fn;
    // wave rendezvous scratch: [wave][parity][lane][bytes]
    alignas(64) unsigned char xbuf[16][2][64][160];
    unsigned xpar[16][64];
};
extern Ctx g;
void launch(std::function<void()> fn, dim3 grid, dim3 block);
void block_barrier();
void wave_barrier();

// deposit `n` bytes for this lane, rendezvous, return pointer to the wave's deposit area (64 x 160 B)
inline unsigned char (*exchange(const void* src, int n))[160] {
    Fiber* f = g.cur;
    unsigned par = g.xpar[f->wave][f->lane]++ & 1;
    memcpy(g.xbuf[f->wave][par][f->lane], src, n);
    wave_barrier();
    return g.xbuf[f->wave][par];
}
}  // namespace emu

#define threadIdx (emu::g.cur->tid)
#define blockIdx (emu::g.bIdx)
#define blockDim (emu::g.bDim)
#define gridDim (emu::g.gDim)

inline void __syncthreads() { emu::block_barrier(); }

template <class T> inline T __shfl_xor(T v, int mask, int width = 64) {
    auto buf = emu::exchange(&v, sizeof(T));
    T r;
    memcpy(&r, buf[(emu::g.cur->lane ^ mask) & 63], sizeof(T));
    return r;
}
template <class T> inline T __shfl(T v, int src, int width = 64) {
    auto buf = emu::exchange(&v, sizeof(T));
    T r;
    memcpy(&r, buf[src & 63], sizeof(T));
    return r;
}

inline float atomicAdd(float* p, float v) { float o = *p; *p = o + v; return o; }
inline int atomicAdd(int* p, int v) { int o = *p; *p = o + v; return o; }
inline unsigned atomicAdd(unsigned* p, unsigned v) { unsigned o = *p; *p = o + v; return o; }

inline float __expf(float x) { return expf(x); }
inline float __frcp_rn(float x) { return 1.0f / x; }
inline unsigned __umulhi(unsigned a, unsigned b) { return (unsigned)(((unsigned long long)a * b) >> 32); }

namespace emu {
template <class K, class... A> inline void launch_k(K k, dim3 gr, dim3 bl, A... a) {
    launch([=]() { k(a...); }, gr, bl);
}
}
#define hipLaunchKernelGGL(kern, grid, block, shm, stream, ...) emu::launch_k(kern, (grid), (block), __VA_ARGS__)
