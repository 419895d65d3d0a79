// Host-side helpers shared by the capi_*.hip translation units (argument checks, launch geometry, the tuning record).
#pragma once
#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include <type_traits>

#include "common.hpp"
#include "../../include/rvt_hip.h"

namespace rvt {
// The process-wide tuning / routing record (include/rvt_hip.h: RvtTuning, rvt_set_tuning).  Defined in capi_core.hip.
static inline const RvtTuning& tuning() { return g_tuning; }

static inline int pow2_ge(int v) { int p = 1; while (p < v) p <<= 1; return p; }
static inline int imin(int a, int b) { return a < b ? a : b; }
static inline int imax(int a, int b) { return a > b ? a : b; }
static inline int grid_for(size_t units, int cap = 2048) {
    size_t g = (units + 255) / 256;
    if (g < 1) g = 1;
    return (int)(g > (size_t)cap ? cap : g);
}
// one workgroup per CU (stem kernels, chain-MLP weight gradient); tuning.one_per_cu_grid: a smaller grid (tests: multi-tile walks)
static inline int one_per_cu_grid(int n_tiles) {
    const int cap = tuning().one_per_cu_grid > 0 ? tuning().one_per_cu_grid : 256;
    return n_tiles < 1 ? 1 : (n_tiles < cap ? n_tiles : cap);
}
}  // namespace rvt

// persistent grid = exactly the workgroups the chip holds at once for THIS kernel instantiation (registers + LDS)
// resident workgroups per CU of a kernel, queried once per kernel (the occupancy API is not free and must not run per
// launch; keyed by the kernel's address because several instantiations share one function type)
template <class K> static int resident_per_cu(K kernel, int threads, int fallback) {
#ifdef RVT_EMU
    return fallback;
#else
    struct Entry { const void* k; int v; };
    static Entry cache[64];
    static int n = 0;
    const void* key = reinterpret_cast<const void*>(kernel);
    for (int i = 0; i < n; i++) if (cache[i].k == key) return cache[i].v;
    int nb = 0, v = fallback;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kernel, threads, 0) == hipSuccess && nb > 0) v = nb;
    if (n < 64) cache[n++] = Entry{key, v};
    return v;
#endif
}

#define DISPATCH_DTYPE(dtype, ...)                                   \
    do {                                                             \
        if ((dtype) == RVT_F32) { typedef float T; __VA_ARGS__; }    \
        else if ((dtype) == RVT_BF16) { typedef bf16 T; __VA_ARGS__; } \
        else { set_last_error("bad dtype %d", (int)(dtype)); return 1; } \
    } while (0)

#define DISPATCH_BN(N, ...)                                          \
    do {                                                             \
        if ((N) <= 64) { constexpr int BN = 64; __VA_ARGS__; }       \
        else { constexpr int BN = 128; __VA_ARGS__; }                \
    } while (0)
