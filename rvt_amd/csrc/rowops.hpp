// Row-wise / element-wise kernels of the RVT hot path: LayerNorm fwd/bwd, column sums (bias
// gradients), the event-tensor prepack (uint8/float NCHW -> padded channels-last T), the ConvLSTM
// gate backward.  All are HBM-bound: every thread moves 16-byte (bf16) / 32-byte (f32) vectors.
#pragma once
#include "common.hpp"

namespace rvt {

// reduce over groups of G consecutive lanes (G power of two <= 64)
__device__ __forceinline__ float group_sum(float v, int G) {
    for (int m = 1; m < G; m <<= 1) v += __shfl_xor(v, m);
    return v;
}

// ---- LayerNorm forward over the channel axis (reference maxvit.py:172,177,229,241; eps 1e-5) ----
// G lanes share a row (G = pow2 >= C/8), each lane holds 8 channels; 64/G rows per wave.
template <class T>
__global__ void __launch_bounds__(256)
ln_fwd_kernel(const T* __restrict__ x, const float* __restrict__ w, const float* __restrict__ b, T* __restrict__ y,
              int rows, int C, int G, float eps) {
    const int lane = threadIdx.x & 63;
    const int rpw = 64 / G;
    const int cl = lane % G;
    const bool cvalid = cl * 8 < C;
    const int wave_global = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int n_waves = gridDim.x * 4;
    float wv[8], bv[8];
#pragma unroll
    for (int i = 0; i < 8; i++) { wv[i] = cvalid ? w[cl * 8 + i] : 0.f; bv[i] = cvalid ? b[cl * 8 + i] : 0.f; }
    const int n_iter = (rows + rpw * n_waves - 1) / (rpw * n_waves);
    for (int it = 0; it < n_iter; it++) {
        const int row = (it * n_waves + wave_global) * rpw + lane / G;
        const bool valid = cvalid && row < rows;
        float v[8];
#pragma unroll
        for (int i = 0; i < 8; i++) v[i] = 0.f;
        if (valid) frag_to_float<T>(frag_load<T>(x + (size_t)row * C + cl * 8), v);
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 8; i++) s += v[i];
        const float mean = group_sum(s, G) / (float)C;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < 8; i++) { float d = valid ? v[i] - mean : 0.f; q += d * d; }
        const float rstd = 1.0f / sqrtf(group_sum(q, G) / (float)C + eps);
        if (valid) {
            float o[8];
#pragma unroll
            for (int i = 0; i < 8; i++) o[i] = (v[i] - mean) * rstd * wv[i] + bv[i];
            frag_store<T>(y + (size_t)row * C + cl * 8, frag_from_float<T>(o));
        }
    }
}

// ---- LayerNorm backward ------------------------------------------------------------------------
// dx = rstd * (g - mean(g) - xhat*mean(g*xhat)),  g = dy*w;   dx_out = dx (+ dres);  dw += dy*xhat, db += dy
template <class T>
__global__ void __launch_bounds__(256)
ln_bwd_kernel(const T* __restrict__ x, const float* __restrict__ w, const T* __restrict__ dy, const T* __restrict__ dres,
              T* __restrict__ dx, float* __restrict__ dw, float* __restrict__ db, int rows, int C, int G, float eps) {
    const int lane = threadIdx.x & 63;
    const int rpw = 64 / G;
    const int cl = lane % G;
    const bool cvalid = cl * 8 < C;
    const int wave_global = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int n_waves = gridDim.x * 4;
    float wv[8], aw[8], ab[8];
#pragma unroll
    for (int i = 0; i < 8; i++) { wv[i] = cvalid ? w[cl * 8 + i] : 0.f; aw[i] = 0.f; ab[i] = 0.f; }
    const int n_iter = (rows + rpw * n_waves - 1) / (rpw * n_waves);
    // two row groups per trip: all loads of both are issued before any arithmetic, and ahead of the previous pair's
    // stores being waited for (bytes in flight per wave are what bounds this kernel)
    for (int it = 0; it < n_iter; it += 2) {
        int row[2]; bool valid[2];
        frag_t<T> fx[2], fd[2], fr[2];
#pragma unroll
        for (int u = 0; u < 2; u++) {
            row[u] = ((it + u) * n_waves + wave_global) * rpw + lane / G;
            valid[u] = cvalid && (it + u) < n_iter && row[u] < rows;
            const size_t o = valid[u] ? (size_t)row[u] * C + cl * 8 : 0;
            fx[u] = frag_load<T>(x + o);
            fd[u] = frag_load<T>(dy + o);
            fr[u] = dres ? frag_load<T>(dres + o) : frag_zero<T>();
        }
#pragma unroll
        for (int u = 0; u < 2; u++) {
            float v[8], d[8], r[8];
            frag_to_float<T>(fx[u], v); frag_to_float<T>(fd[u], d); frag_to_float<T>(fr[u], r);
#pragma unroll
            for (int i = 0; i < 8; i++) { v[i] = valid[u] ? v[i] : 0.f; d[i] = valid[u] ? d[i] : 0.f; }
            float s = 0.f;
#pragma unroll
            for (int i = 0; i < 8; i++) s += v[i];
            const float mean = group_sum(s, G) / (float)C;
            float q = 0.f;
#pragma unroll
            for (int i = 0; i < 8; i++) { float t = valid[u] ? v[i] - mean : 0.f; q += t * t; }
            const float rstd = 1.0f / sqrtf(group_sum(q, G) / (float)C + eps);
            float xh[8], gsum = 0.f, gxsum = 0.f;
#pragma unroll
            for (int i = 0; i < 8; i++) {
                xh[i] = valid[u] ? (v[i] - mean) * rstd : 0.f;
                float g = d[i] * wv[i];
                gsum += g; gxsum += g * xh[i];
                aw[i] += d[i] * xh[i]; ab[i] += d[i];
            }
            const float m1 = group_sum(gsum, G) / (float)C;
            const float m2 = group_sum(gxsum, G) / (float)C;
            if (valid[u]) {
                float o[8];
#pragma unroll
                for (int i = 0; i < 8; i++) o[i] = rstd * (d[i] * wv[i] - m1 - xh[i] * m2) + r[i];
                frag_store<T>(dx + (size_t)row[u] * C + cl * 8, frag_from_float<T>(o));
            }
        }
    }
    // fold the 64/G row-lanes of a wave that own the same columns, then the 4 waves through LDS:
    // one atomic per column per WORKGROUP (same-address atomics serialise at the L2)
    __shared__ float red[4][2][512];
#pragma unroll
    for (int i = 0; i < 8; i++) {
        for (int m = G; m < 64; m <<= 1) { aw[i] += __shfl_xor(aw[i], m); ab[i] += __shfl_xor(ab[i], m); }
    }
    const int wave = threadIdx.x >> 6;
    if (cvalid && lane < G) {
#pragma unroll
        for (int i = 0; i < 8; i++) { red[wave][0][cl * 8 + i] = aw[i]; red[wave][1][cl * 8 + i] = ab[i]; }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) {
        atomicAdd(dw + c, red[0][0][c] + red[1][0][c] + red[2][0][c] + red[3][0][c]);
        atomicAdd(db + c, red[0][1][c] + red[1][1][c] + red[2][1][c] + red[3][1][c]);
    }
}

// ---- column sums: out[n] += sum_m x[m][n]   (bias / LayerScale gradients) ------------------------
// NCP = pow2 >= N/8 (<= 256) threads own one 8-column chunk each; 256/NCP row lanes per block.
template <class T>
__global__ void __launch_bounds__(256)
colsum_kernel(const T* __restrict__ x, float* __restrict__ out, int rows, int N, int NCP) {
    __shared__ float red[256 * 8];
    const int tid = threadIdx.x;
    const int c = tid % NCP + blockIdx.y * NCP;
    const int rl = tid / NCP, nrl = 256 / NCP;
    const bool cvalid = c * 8 < N;
    float a[8];
#pragma unroll
    for (int i = 0; i < 8; i++) a[i] = 0.f;
    if (cvalid) {
        for (int row = blockIdx.x * nrl + rl; row < rows; row += gridDim.x * nrl) {
            float v[8]; frag_to_float<T>(frag_load<T>(x + (size_t)row * N + c * 8), v);
#pragma unroll
            for (int i = 0; i < 8; i++) a[i] += v[i];
        }
    }
#pragma unroll
    for (int i = 0; i < 8; i++) red[tid * 8 + i] = a[i];
    __syncthreads();
    if (rl == 0 && cvalid) {
        for (int r = 1; r < nrl; r++)
#pragma unroll
            for (int i = 0; i < 8; i++) a[i] += red[(r * NCP + tid) * 8 + i];
#pragma unroll
        for (int i = 0; i < 8; i++) atomicAdd(out + c * 8 + i, a[i]);
    }
}

// ---- event-tensor prepack (reference modules/detection.py:133-134 cast + utils/padding.py:29-44) ----
// src: [F][Cin][h][w] uint8 or float, unpadded.  dst: [F][H][W][Cp] T, zero padded bottom/right and in channels.
// A workgroup transposes one piece (PrepackSeg pixels) of an image row through LDS: plane-major source rows come in as
// whole contiguous segments (uint8 as 4-byte words when the geometry allows), channel-last pixels go out as
// consecutive 16-byte chunks — both sides of the transpose touch memory in full cache lines.
// pixels per work item: a whole 640-pixel row for uint8 planes (20 KiB of LDS; measured 1.58 ms vs 2.30 at 128 pixels on the
// 1 Mpx batch), 128 for float planes (LDS)
template <class S> struct PrepackSeg { static constexpr int value = sizeof(S) == 1 ? 640 : 128; };
template <class T, class S>
__global__ void __launch_bounds__(256)
prepack_kernel(const S* __restrict__ src, T* __restrict__ dst, int F, int Cin, int h, int w, int H, int W, int Cp) {
    constexpr int SEG = PrepackSeg<S>::value;
    constexpr int CMAX = 32;                                  // staged channels (host checks Cin <= CMAX)
    __shared__ S plane[CMAX][SEG + 4];
    const int tid = threadIdx.x;
    const int segs = (W + SEG - 1) / SEG;
    const int cpp = Cp / 8;
    const size_t n_items = (size_t)F * H * segs;
    for (size_t item = blockIdx.x; item < n_items; item += gridDim.x) {
        const int sg = (int)(item % segs);
        const int y = (int)((item / segs) % H);
        const int f = (int)(item / ((size_t)segs * H));
        const int x0 = sg * SEG;
        const int nx = (W - x0) < SEG ? (W - x0) : SEG;       // destination pixels of this piece
        const int nsrc = y < h ? ((w - x0) < 0 ? 0 : ((w - x0) < SEG ? (w - x0) : SEG)) : 0;   // real source pixels
        if (nsrc > 0) {
            const S* row0 = src + (((size_t)f * Cin) * h + y) * w + x0;
            const size_t pstride = (size_t)h * w;
            if (sizeof(S) == 1 && (w % 4) == 0 && (nsrc % 4) == 0) {
                const int wpr = nsrc / 4;                     // 4-byte words per plane row piece
                for (int e = tid; e < Cin * wpr; e += 256) {
                    const int c = e / wpr, q = e % wpr;
                    *reinterpret_cast<uint32_t*>(&plane[c][q * 4]) =
                        *reinterpret_cast<const uint32_t*>(row0 + c * pstride + q * 4);
                }
            } else {
                for (int e = tid; e < Cin * nsrc; e += 256) {
                    const int c = e / nsrc, q = e % nsrc;
                    plane[c][q] = row0[c * pstride + q];
                }
            }
        }
        __syncthreads();
        T* drow = dst + (((size_t)f * H + y) * W + x0) * Cp;
        for (int u = tid; u < nx * cpp; u += 256) {
            const int px = u / cpp, c0 = (u % cpp) * 8;
            float v[8];
#pragma unroll
            for (int i = 0; i < 8; i++) v[i] = (px < nsrc && c0 + i < Cin) ? (float)plane[c0 + i][px] : 0.f;
            frag_store<T>(drow + (size_t)px * Cp + c0, frag_from_float<T>(v));
        }
        __syncthreads();
    }
}

// ---- ConvLSTM gate backward (element-wise part of BPTT; reference forward rnn.py:57-67) --------
// dh = dh_in (+ dh_rec);  do = dh*tanh(c);  dc = dc_rec + dh*o*(1-tanh(c)^2);
// dz = [dc*c_prev*f(1-f), dc*g*i(1-i), do*o(1-o), dc*i*(1-g^2)];  dc_rec <- dc*f
template <class T>
__global__ void __launch_bounds__(256)
lstm_gates_bwd_kernel(const T* __restrict__ dh_in, const T* __restrict__ dh_rec, float* __restrict__ dc_rec,
                      const T* __restrict__ gates, const float* __restrict__ c_new, const float* __restrict__ c_prev,
                      T* __restrict__ dz, int M, int C) {
    const int cpr = C / 8;
    const size_t total = (size_t)M * cpr;
    for (size_t u = (size_t)blockIdx.x * 256 + threadIdx.x; u < total; u += (size_t)gridDim.x * 256) {
        const size_t m = u / cpr;
        const int c0 = (int)(u % cpr) * 8;
        float dh[8], f[8], ig[8], o[8], g[8];
        frag_to_float<T>(frag_load<T>(dh_in + m * C + c0), dh);
        if (dh_rec) {
            float r[8]; frag_to_float<T>(frag_load<T>(dh_rec + m * C + c0), r);
#pragma unroll
            for (int i = 0; i < 8; i++) dh[i] += r[i];
        }
        const T* gp = gates + m * 4 * C + c0;
        frag_to_float<T>(frag_load<T>(gp), f);
        frag_to_float<T>(frag_load<T>(gp + C), ig);
        frag_to_float<T>(frag_load<T>(gp + 2 * C), o);
        frag_to_float<T>(frag_load<T>(gp + 3 * C), g);
        float zf[8], zi[8], zo[8], zg[8];
        float* dcp = dc_rec + m * C + c0;
        const float* cn = c_new + m * C + c0;
        const float* cp = c_prev + m * C + c0;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            float tc = tanh_f(cn[i]);
            float dc = dcp[i] + dh[i] * o[i] * (1.f - tc * tc);
            zo[i] = dh[i] * tc * o[i] * (1.f - o[i]);
            zf[i] = dc * cp[i] * f[i] * (1.f - f[i]);
            zi[i] = dc * g[i] * ig[i] * (1.f - ig[i]);
            zg[i] = dc * ig[i] * (1.f - g[i] * g[i]);
            dcp[i] = dc * f[i];
        }
        T* zp = dz + m * 4 * C + c0;
        frag_store<T>(zp, frag_from_float<T>(zf));
        frag_store<T>(zp + C, frag_from_float<T>(zi));
        frag_store<T>(zp + 2 * C, frag_from_float<T>(zo));
        frag_store<T>(zp + 3 * C, frag_from_float<T>(zg));
    }
}

// zero the state rows of the samples flagged in `mask` (reference modules/utils/detection.py:96-113)
template <class S>
__global__ void __launch_bounds__(256)
state_reset_kernel(S* __restrict__ st, const unsigned char* __restrict__ mask, int B, size_t per_sample) {
    const size_t total = (size_t)B * per_sample;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256)
        if (mask[i / per_sample]) st[i] = (S)0.0f;
}

// ---- labelled-frame gather (reference modules/utils/detection.py:32-46, BackboneFeatureSelector): the feature maps of the
// (t, b) frames that carry labels, concatenated — dst[n] = src[idx[n]] over whole frames of `frame_vec` 16-byte vectors;
// SCATTER: the backward, dst[idx[n]] = src[n] into a zero-filled gradient (indices are distinct by construction).
template <bool SCATTER>
__global__ void __launch_bounds__(256)
gather_frames_kernel(const u32x4* __restrict__ src, const int* __restrict__ idx, u32x4* __restrict__ dst, int n_sel, size_t frame_vec) {
    const size_t total = (size_t)n_sel * frame_vec;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const size_t n = i / frame_vec, v = i - n * frame_vec;
        const size_t f = (size_t)idx[n];
        if (SCATTER) dst[f * frame_vec + v] = src[i];
        else dst[i] = src[f * frame_vec + v];
    }
}

}  // namespace rvt

namespace rvt {

// ---- depth-wise k x k convolution on channels-last maps (reference rnn.py:25-29,50-54: groups = channels,
// padding k/2, stride 1, with bias).  Not GEMM-shaped (one input channel per output channel): HBM/L2-bound
// stencil; a thread owns 8 channels of one pixel and walks the k*k taps with 16-byte loads.
//   FLIP=false: y = dwconv(x) + b           FLIP=true: dx = dwconv^T(dy) (taps mirrored, no bias)
// x/y are [N][H][W][ld] slices starting at channel offset 0 with Cg channels used (ld >= Cg).
template <class T, bool FLIP>
__global__ void __launch_bounds__(256)
dwconv_kernel(const T* __restrict__ x, int ldx, const float* __restrict__ w, const float* __restrict__ b,
              T* __restrict__ y, int ldy, int N, int H, int W, int Cg, int k) {
    const int cpr = Cg / 8;
    const int pad = k / 2;
    const size_t total = (size_t)N * H * W * cpr;
    for (size_t u = (size_t)blockIdx.x * 256 + threadIdx.x; u < total; u += (size_t)gridDim.x * 256) {
        const int c0 = (int)(u % cpr) * 8;
        const size_t pix = u / cpr;
        const int xx = (int)(pix % W);
        const int yy = (int)((pix / W) % H);
        const size_t n = pix / ((size_t)W * H);
        float acc[8];
#pragma unroll
        for (int i = 0; i < 8; i++) acc[i] = (b && !FLIP) ? b[c0 + i] : 0.f;
        for (int ky = 0; ky < k; ky++) {
            const int iy = yy + ky - pad;
            if (iy < 0 || iy >= H) continue;
            for (int kx = 0; kx < k; kx++) {
                const int ix = xx + kx - pad;
                if (ix < 0 || ix >= W) continue;
                float v[8];
                frag_to_float<T>(frag_load<T>(x + ((n * H + iy) * W + ix) * ldx + c0), v);
                const int tap = FLIP ? (k - 1 - ky) * k + (k - 1 - kx) : ky * k + kx;
#pragma unroll
                for (int i = 0; i < 8; i++) acc[i] += v[i] * w[(size_t)(c0 + i) * k * k + tap];
            }
        }
        frag_store<T>(y + pix * ldy + c0, frag_from_float<T>(acc));
    }
}

// dw[c][ky][kx] += sum_pixels x[pix + (ky,kx) - pad][c] * dy[pix][c];  db[c] += sum dy[pix][c]
// One workgroup walks a strip of pixels; a thread owns (8 channels) and accumulates all k*k taps (k <= 3 -> 9x8 regs);
// the 256/cpr pixel-lanes of a workgroup are folded through LDS, then one atomic per (channel, tap) per workgroup.
template <class T>
__global__ void __launch_bounds__(256)
dwconv_wgrad_kernel(const T* __restrict__ x, int ldx, const T* __restrict__ dy, int ldy, float* __restrict__ dw,
                    float* __restrict__ db, int N, int H, int W, int Cg, int k, int CP) {
    // CP = pow2 >= Cg/8 (<= 256): threads tid%CP own a channel chunk, tid/CP are pixel lanes
    __shared__ float red[256][8];
    const int tid = threadIdx.x;
    const int cc = tid % CP + blockIdx.y * CP;
    const int pl = tid / CP, npl = 256 / CP;
    const bool cvalid = cc * 8 < Cg;
    const int c0 = cc * 8;
    const int pad = k / 2;
    const size_t npix = (size_t)N * H * W;
    float aw[9][8], ab[8];
#pragma unroll
    for (int t = 0; t < 9; t++)
#pragma unroll
        for (int i = 0; i < 8; i++) aw[t][i] = 0.f;
#pragma unroll
    for (int i = 0; i < 8; i++) ab[i] = 0.f;
    if (cvalid) {
        for (size_t pix = (size_t)blockIdx.x * npl + pl; pix < npix; pix += (size_t)gridDim.x * npl) {
            const int xx = (int)(pix % W);
            const int yy = (int)((pix / W) % H);
            const size_t n = pix / ((size_t)W * H);
            float g[8];
            frag_to_float<T>(frag_load<T>(dy + pix * ldy + c0), g);
#pragma unroll
            for (int i = 0; i < 8; i++) ab[i] += g[i];
#pragma unroll
            for (int ky = 0; ky < 3; ky++) {
                if (ky >= k) continue;
                const int iy = yy + ky - pad;
                if (iy < 0 || iy >= H) continue;
#pragma unroll
                for (int kx = 0; kx < 3; kx++) {
                    if (kx >= k) continue;
                    const int ix = xx + kx - pad;
                    if (ix < 0 || ix >= W) continue;
                    float v[8];
                    frag_to_float<T>(frag_load<T>(x + ((n * H + iy) * W + ix) * ldx + c0), v);
#pragma unroll
                    for (int i = 0; i < 8; i++) aw[ky * 3 + kx][i] += v[i] * g[i];
                }
            }
        }
    }
    for (int t = 0; t <= k * k; t++) {           // t == k*k: the bias sums
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 8; i++) {
            float v = 0.f;
            if (t == k * k) v = ab[i];
            else {
#pragma unroll
                for (int q = 0; q < 9; q++) if (q == (t / k) * 3 + (t % k)) v = aw[q][i];
            }
            red[tid][i] = v;
        }
        __syncthreads();
        if (pl == 0 && cvalid) {
#pragma unroll
            for (int i = 0; i < 8; i++) {
                float s = 0.f;
                for (int r = 0; r < npl; r++) s += red[r * CP + tid][i];
                if (t == k * k) atomicAdd(db + c0 + i, s);
                else atomicAdd(dw + (size_t)(c0 + i) * k * k + t, s);
            }
        }
    }
}

}  // namespace rvt

namespace rvt {

// ---- token masking (reference maxvit_rnn.py:174-176: x[token_mask] = mask_token) -------------------------------
// forward: rows flagged in mask[M] are overwritten with the (fp32) mask token.
template <class T>
__global__ void __launch_bounds__(256)
token_mask_fwd_kernel(T* __restrict__ x, const unsigned char* __restrict__ mask, const float* __restrict__ token, int M, int C) {
    const int cpr = C / 8;
    const size_t total = (size_t)M * cpr;
    for (size_t u = (size_t)blockIdx.x * 256 + threadIdx.x; u < total; u += (size_t)gridDim.x * 256) {
        const size_t m = u / cpr;
        if (!mask[m]) continue;
        const int c0 = (int)(u % cpr) * 8;
        float v[8];
#pragma unroll
        for (int i = 0; i < 8; i++) v[i] = token[c0 + i];
        frag_store<T>(x + m * C + c0, frag_from_float<T>(v));
    }
}

// backward: dtoken[c] += sum over masked rows of dx[m][c]; dx of masked rows is zeroed (the conv path gets no gradient there)
template <class T>
__global__ void __launch_bounds__(256)
token_mask_bwd_kernel(T* __restrict__ dx, const unsigned char* __restrict__ mask, float* __restrict__ dtoken, int M, int C, int NCP) {
    __shared__ float red[256 * 8];
    const int tid = threadIdx.x;
    const int c = tid % NCP + blockIdx.y * NCP;
    const int rl = tid / NCP, nrl = 256 / NCP;
    const bool cvalid = c * 8 < C;
    float a[8];
#pragma unroll
    for (int i = 0; i < 8; i++) a[i] = 0.f;
    if (cvalid) {
        for (int row = blockIdx.x * nrl + rl; row < M; row += gridDim.x * nrl) {
            if (!mask[row]) continue;
            T* p = dx + (size_t)row * C + c * 8;
            float v[8]; frag_to_float<T>(frag_load<T>(p), v);
#pragma unroll
            for (int i = 0; i < 8; i++) a[i] += v[i];
            frag_store<T>(p, frag_zero<T>());
        }
    }
#pragma unroll
    for (int i = 0; i < 8; i++) red[tid * 8 + i] = a[i];
    __syncthreads();
    if (rl == 0 && cvalid) {
        for (int r = 1; r < nrl; r++)
#pragma unroll
            for (int i = 0; i < 8; i++) a[i] += red[(r * NCP + tid) * 8 + i];
#pragma unroll
        for (int i = 0; i < 8; i++) atomicAdd(dtoken + c * 8 + i, a[i]);
    }
}

}  // namespace rvt
