// Event stream -> stacked histogram (reference data/utils/representations.py:76-117, StackedHistogram.construct):
// the producer of the 20-channel uint8 event tensors the backbone consumes ("next" row f4 of SURVEY.md §8).
//
//   t_idx = min(floor( float32(t - t[0]) / float32(max(t[n-1] - t[0], 1)) * bins ), bins - 1)     (float32 like torch)
//   rep[pol][t_idx][y][x] += 1      (uint8 accumulation in fastmode: wraps modulo 256; int16 otherwise)
//   rep = clamp(rep, 0, count_cutoff) -> uint8, viewed as (2*bins, H, W)
//
// An HBM-bound scatter: 32 bytes of event record in, one 4-byte atomic per event on a 2*bins*H*W counter image
// (18 MB at 1 Mpx: L2 / Infinity-Cache resident), then one pass that folds the reference's wrap-around and clamp and
// narrows to uint8.  Integer work: results are bit-identical to the reference for any event order.
#pragma once
#include "common.hpp"

namespace rvt {

__global__ void __launch_bounds__(256)
hist_count_kernel(const long long* __restrict__ x, const long long* __restrict__ y, const long long* __restrict__ pol,
                  const long long* __restrict__ time, size_t n, int bins, int H, int W, unsigned* __restrict__ counts) {
    const long long t0 = time[0], t1 = time[n - 1];
    const float den = (float)((t1 - t0) > 1 ? (t1 - t0) : 1);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const long long xi = x[i], yi = y[i], pi = pol[i];
        float tn = (float)(time[i] - t0) / den;          // correctly rounded fp32 division, as torch's
        tn = tn * (float)bins;
        int ti = (int)floorf(tn);
        ti = ti < bins - 1 ? ti : bins - 1;
        if (xi < 0 || xi >= W || yi < 0 || yi >= H || pi < 0 || pi > 1 || ti < 0) continue;   // (reference: index error)
        atomicAdd(counts + (((size_t)pi * bins + ti) * H + yi) * W + xi, 1u);
    }
}

__global__ void __launch_bounds__(256)
hist_finalize_kernel(const unsigned* __restrict__ counts, unsigned char* __restrict__ out, size_t n, int cutoff, int fastmode) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const unsigned c = counts[i];
        int v;
        if (fastmode) v = (int)(c & 0xffu);                            // uint8 accumulator: modulo 256
        else { v = (int)(short)(c & 0xffffu); v = v < 0 ? 0 : v; }     // int16 accumulator, clamp(min=0)
        out[i] = (unsigned char)(v < cutoff ? v : cutoff);
    }
}

}  // namespace rvt
