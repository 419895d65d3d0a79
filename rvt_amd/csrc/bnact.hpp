// BatchNorm2d + SiLU of the YOLOX PAFPN's BaseConv units on channels-last maps (reference
// models/detection/yolox/models/network_blocks.py:29-53: Conv2d(bias=False) -> BatchNorm2d -> SiLU; nn.BatchNorm2d defaults
// eps 1e-5, momentum 0.1).  SURVEY.md section 8 row f2.  The conv itself runs on the GEMM engine (rvt_conv_fwd / _dgrad / _wgrad);
// these are the row-wise kernels around it, all HBM-bound (one 16-byte vector of 8 channels per thread and row):
//   bn_stats          per-channel sum and sum of squares of the conv output over all N*H*W rows (training: batch statistics;
//                     under data parallelism the two vectors are what SyncBatchNorm all-reduces, train.py:133)
//   bn_finalize       mean / rstd, the fused scale = gamma * rstd and shift = beta - mean * scale, running-statistics update
//   bn_act_fwd        y = silu(x * scale + shift)
//   bn_act_bwd_stats  dz = dy * silu'(z), z = x * scale + shift;  sum(dz), sum(dz * xhat) per channel  (= dbeta, dgamma)
//   bn_act_bwd_apply  dx = gamma * rstd * (dz - sum(dz) / N - xhat * sum(dz * xhat) / N)
#pragma once
#include "common.hpp"

namespace rvt {

enum { BN_ACT_NONE = 0, BN_ACT_SILU = 1 };

__device__ __forceinline__ float silu_f(float z) { return z * sigmoid_f(z); }
__device__ __forceinline__ float silu_grad_f(float z) { const float s = sigmoid_f(z); return s * (1.f + z * (1.f - s)); }

// threads: Gp = pow2 >= C/8 column groups x 256/Gp row lanes; a workgroup strides over row blocks
template <class T, int NACC, class F>
__device__ __forceinline__ void bn_column_reduce(int rows, int C, int Gp, float* const* outs, F&& per_row) {
    const int tid = threadIdx.x, cg = tid % Gp, r0 = tid / Gp, nrl = 256 / Gp;
    const bool cvalid = cg * 8 < C;
    float acc[NACC][8];
#pragma unroll
    for (int a = 0; a < NACC; a++)
#pragma unroll
        for (int i = 0; i < 8; i++) acc[a][i] = 0.f;
    if (cvalid)
        for (int row = blockIdx.x * nrl + r0; row < rows; row += gridDim.x * nrl) per_row(row, cg, acc);
    __shared__ float red[256 * 8];                          // [row lane][Gp * 8 columns], one accumulator at a time
#pragma unroll
    for (int a = 0; a < NACC; a++) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 8; i++) red[r0 * (Gp * 8) + cg * 8 + i] = acc[a][i];
        __syncthreads();
        for (int c = tid; c < C; c += 256) {
            float s = 0.f;
            for (int r = 0; r < nrl; r++) s += red[r * (Gp * 8) + c];
            atomicAdd(outs[a] + c, s);                      // one atomic per column per workgroup
        }
    }
}

template <class T>
__global__ void __launch_bounds__(256)
bn_stats_kernel(const T* __restrict__ x, float* __restrict__ sum, float* __restrict__ sumsq, int rows, int C, int Gp) {
    float* outs[2] = {sum, sumsq};
    bn_column_reduce<T, 2>(rows, C, Gp, outs, [&](int row, int cg, float (&acc)[2][8]) {
        float v[8];
        frag_to_float<T>(frag_load<T>(x + (size_t)row * C + cg * 8), v);
#pragma unroll
        for (int i = 0; i < 8; i++) { acc[0][i] += v[i]; acc[1][i] += v[i] * v[i]; }
    });
}

// one workgroup; training: batch statistics (biased variance for the normalisation, unbiased for running_var, as nn.BatchNorm2d)
__global__ void __launch_bounds__(256)
bn_finalize_kernel(const float* __restrict__ sum, const float* __restrict__ sumsq, float count, const float* __restrict__ gamma,
                   const float* __restrict__ beta, float eps, float momentum, float* __restrict__ running_mean,
                   float* __restrict__ running_var, float* __restrict__ mean_out, float* __restrict__ rstd_out,
                   float* __restrict__ scale, float* __restrict__ shift, int C, int training) {
    for (int c = threadIdx.x + blockIdx.x * blockDim.x; c < C; c += blockDim.x * gridDim.x) {
        float mean, var;
        if (training) {
            mean = sum[c] / count;
            var = fmaxf(sumsq[c] / count - mean * mean, 0.f);
            if (running_mean != nullptr) {
                running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mean;
                running_var[c] = (1.f - momentum) * running_var[c] + momentum * var * (count > 1.f ? count / (count - 1.f) : 1.f);
            }
        } else {
            mean = running_mean[c];
            var = running_var[c];
        }
        const float rstd = 1.0f / sqrtf(var + eps);
        const float sc = gamma[c] * rstd;
        if (mean_out != nullptr) { mean_out[c] = mean; rstd_out[c] = rstd; }
        scale[c] = sc;
        shift[c] = beta[c] - mean * sc;
    }
}

// Row-streaming layout of the element-wise kernels: thread = (column group cg = tid % Gp, row lane tid / Gp), Gp = pow2 >= C / 8.
// A thread keeps its 8 channels' coefficients in registers for the whole launch and walks rows with stride gridDim * (256 / Gp):
// one 16-byte load / store per tensor and row, nothing else touches memory.
template <class T>
__global__ void __launch_bounds__(256)
bn_act_fwd_kernel(const T* __restrict__ x, const float* __restrict__ scale, const float* __restrict__ shift, T* __restrict__ y,
                  int rows, int C, int Gp, int act) {
    const int cg = threadIdx.x % Gp, r0 = threadIdx.x / Gp, nrl = 256 / Gp;
    if (cg * 8 >= C) return;
    float sc[8], sh[8];
#pragma unroll
    for (int i = 0; i < 8; i++) { sc[i] = scale[cg * 8 + i]; sh[i] = shift[cg * 8 + i]; }
    for (int row = blockIdx.x * nrl + r0; row < rows; row += gridDim.x * nrl) {
        float v[8], o[8];
        frag_to_float<T>(frag_load<T>(x + (size_t)row * C + cg * 8), v);
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const float z = fmaf(v[i], sc[i], sh[i]);
            o[i] = act == BN_ACT_SILU ? silu_f(z) : z;
        }
        frag_store<T>(y + (size_t)row * C + cg * 8, frag_from_float<T>(o));
    }
}

// training forward in one launch: every thread finalizes ITS 8 channels from the (already all-reduced) sums — mean, rstd, scale, shift —
// and streams rows; the first row lane of workgroup 0 also publishes them (the backward needs them) and updates the running statistics
// (what bn_finalize_kernel does as a launch of its own)
template <class T>
__global__ void __launch_bounds__(256)
bn_train_act_fwd_kernel(const T* __restrict__ x, const float* __restrict__ sum, const float* __restrict__ sumsq, float count,
                        const float* __restrict__ gamma, const float* __restrict__ beta, float eps, float momentum,
                        float* __restrict__ running_mean, float* __restrict__ running_var, float* __restrict__ mean_out,
                        float* __restrict__ rstd_out, float* __restrict__ scale_out, float* __restrict__ shift_out, T* __restrict__ y,
                        int rows, int C, int Gp, int act) {
    const int cg = threadIdx.x % Gp, r0 = threadIdx.x / Gp, nrl = 256 / Gp;
    if (cg * 8 >= C) return;
    float sc[8], sh[8];
    const bool publish = blockIdx.x == 0 && r0 == 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const int c = cg * 8 + i;
        const float mean = sum[c] / count;
        const float var = fmaxf(sumsq[c] / count - mean * mean, 0.f);
        const float rstd = 1.0f / sqrtf(var + eps);
        sc[i] = gamma[c] * rstd;
        sh[i] = beta[c] - mean * sc[i];
        if (publish) {
            mean_out[c] = mean; rstd_out[c] = rstd; scale_out[c] = sc[i]; shift_out[c] = sh[i];
            if (running_mean != nullptr) {
                running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mean;
                running_var[c] = (1.f - momentum) * running_var[c] + momentum * var * (count > 1.f ? count / (count - 1.f) : 1.f);
            }
        }
    }
    for (int row = blockIdx.x * nrl + r0; row < rows; row += gridDim.x * nrl) {
        float v[8], o[8];
        frag_to_float<T>(frag_load<T>(x + (size_t)row * C + cg * 8), v);
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const float z = fmaf(v[i], sc[i], sh[i]);
            o[i] = act == BN_ACT_SILU ? silu_f(z) : z;
        }
        frag_store<T>(y + (size_t)row * C + cg * 8, frag_from_float<T>(o));
    }
}

template <class T>
__global__ void __launch_bounds__(256)
bn_act_bwd_stats_kernel(const T* __restrict__ dy, const T* __restrict__ x, const float* __restrict__ scale,
                        const float* __restrict__ shift, const float* __restrict__ mean, const float* __restrict__ rstd,
                        float* __restrict__ dsum, float* __restrict__ dxsum, int rows, int C, int Gp, int act) {
    float* outs[2] = {dsum, dxsum};
    const int cg0 = threadIdx.x % Gp;
    float sc[8], sh[8], mu[8], rs[8];
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const int c = cg0 * 8 + i < C ? cg0 * 8 + i : 0;
        sc[i] = scale[c]; sh[i] = shift[c]; mu[i] = mean[c]; rs[i] = rstd[c];
    }
    bn_column_reduce<T, 2>(rows, C, Gp, outs, [&](int row, int cg, float (&acc)[2][8]) {
        float v[8], d[8];
        frag_to_float<T>(frag_load<T>(x + (size_t)row * C + cg * 8), v);
        frag_to_float<T>(frag_load<T>(dy + (size_t)row * C + cg * 8), d);
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const float z = fmaf(v[i], sc[i], sh[i]);
            const float dz = act == BN_ACT_SILU ? d[i] * silu_grad_f(z) : d[i];
            acc[0][i] += dz;
            acc[1][i] += dz * (v[i] - mu[i]) * rs[i];
        }
    });
}

template <class T>
__global__ void __launch_bounds__(256)
bn_act_bwd_apply_kernel(const T* __restrict__ dy, const T* __restrict__ x, const float* __restrict__ scale,
                        const float* __restrict__ shift, const float* __restrict__ mean, const float* __restrict__ rstd,
                        const float* __restrict__ dsum, const float* __restrict__ dxsum, T* __restrict__ dx, int rows, int C, int Gp,
                        float inv_count, int act) {
    const int cg = threadIdx.x % Gp, r0 = threadIdx.x / Gp, nrl = 256 / Gp;
    if (cg * 8 >= C) return;
    float sc[8], sh[8], mu[8], rs[8], k1[8], k2[8];
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const int c = cg * 8 + i;
        sc[i] = scale[c]; sh[i] = shift[c]; mu[i] = mean[c]; rs[i] = rstd[c];
        k1[i] = dsum[c] * inv_count; k2[i] = dxsum[c] * inv_count;
    }
    for (int row = blockIdx.x * nrl + r0; row < rows; row += gridDim.x * nrl) {
        float v[8], d[8], o[8];
        frag_to_float<T>(frag_load<T>(x + (size_t)row * C + cg * 8), v);
        frag_to_float<T>(frag_load<T>(dy + (size_t)row * C + cg * 8), d);
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const float z = fmaf(v[i], sc[i], sh[i]);
            const float dz = act == BN_ACT_SILU ? d[i] * silu_grad_f(z) : d[i];
            const float xh = (v[i] - mu[i]) * rs[i];
            o[i] = sc[i] * (dz - k1[i] - xh * k2[i]);                                       // scale = gamma * rstd
        }
        frag_store<T>(dx + (size_t)row * C + cg * 8, frag_from_float<T>(o));
    }
}

}  // namespace rvt
