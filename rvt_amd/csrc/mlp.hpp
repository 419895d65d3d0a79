// Fused MLP half of a PartitionAttentionCl block (reference maxvit.py:269 with the MLP of :100-118):
//
//     xout = xmid + gamma2 * ( GELU( LN2(xmid) W1^T + b1 ) W2^T + b2 )
//
// in ONE kernel per 128-token tile: the normalised tile, the 4C-wide hidden activations and both weight panels
// live in LDS / registers only.  HBM traffic is one read of xmid and one write of xout (2*C*sizeof(T) per token)
// instead of the ~17*C*sizeof(T) of the op-by-op chain (LayerNorm r/w, fc1 r + hidden w, fc2 hidden r + res r + w).
// For C <= 128 (stages 1-2 of RVT, where the unfused chain is HBM-bound) this turns the MLP MFMA-bound:
// per tile 2*(4C/JC)*... MFMAs vs 2*128*C*sizeof(T) bytes.
//
// Structure per workgroup (256 threads = 4 waves as 2x2, persistent over token tiles):
//   1. load the [128][C] tile (16-byte vectors, G=C/8 lanes per row), LayerNorm it in registers (wave shuffles,
//      fp32 statistics), store v2 into a swizzled LDS A operand; the raw tile stays in registers as the residual.
//   2. for each hidden chunk j of JC columns:  stage W1_j -> LDS;  H = GELU(v2 W1_j^T + b1_j) (MFMA, epilogue in
//      registers) -> LDS as the next A operand;  stage W2[:, j] -> LDS;  acc += H W2_j^T.
//   3. acc -> fp32 LDS staging -> out = residual + gamma*(acc + b2), 16-byte stores.
// JC = 128 (bf16) / 64 (f32) so that all operands fit the 160 KiB LDS.
#pragma once
#include "common.hpp"
#include "rowops.hpp"

namespace rvt {

// ---- LDS operand matrix: [rows][K] stored as K/BK sub-tiles of [rows][128 bytes], XOR-swizzled like GEMM tiles ----
template <class T> __device__ __forceinline__ char* opm_subtile(char* base, int rows, int kt) { return base + (size_t)kt * rows * 128; }

// element address (for scattered 1-element writes from accumulator layout)
template <class T> __device__ __forceinline__ T* opm_elem(char* base, int rows, int row, int kcol) {
    constexpr int BK = TileGeom<T>::BK;
    const int kt = kcol / BK, kin = kcol % BK;
    const int byte = kin * (int)sizeof(T);
    return reinterpret_cast<T*>(opm_subtile<T>(base, rows, kt) + lds_chunk_off(row, byte >> 4) + (byte & 15));
}

// stage a row-major global block [rows][kcols] (leading dimension ld elements) into an operand matrix; all 256 threads
template <class T> __device__ __forceinline__ void opm_stage(char* base, const T* g, int ld, int rows, int kcols, int tid) {
    constexpr int FPR = TileGeom<T>::FPR;
    const int fpr_g = kcols / 8;
    for (int f = tid; f < rows * fpr_g; f += 256) {
        const int row = f / fpr_g, fcg = f % fpr_g;
        tile_store_frag<T>(opm_subtile<T>(base, rows, fcg / FPR), row, fcg % FPR, frag_load<T>(g + (size_t)row * ld + fcg * 8));
    }
}

// acc[i][j] += A[a_row0 + 32 i + .][0..ktot) . B[b_row0 + 32 j + .][0..ktot)
template <class T, int MI, int NJ>
__device__ __forceinline__ void opm_mma(f32x16 (&acc)[MI][NJ], const char* A, int a_rows, int a_row0, const char* B, int b_rows,
                                        int b_row0, int ktot, int lane) {
    constexpr int BK = TileGeom<T>::BK;
    const int li = lane & 31, half = lane >> 5;
    for (int kt = 0; kt < ktot / BK; kt++) {
        const char* At = A + (size_t)kt * a_rows * 128;
        const char* Bt = B + (size_t)kt * b_rows * 128;
#pragma unroll
        for (int ks = 0; ks < BK / 16; ks++) {
            const int fc = ks * 2 + half;
            frag_t<T> a[MI], b[NJ];
#pragma unroll
            for (int i = 0; i < MI; i++) a[i] = tile_load_frag<T>(At, a_row0 + i * 32 + li, fc);
#pragma unroll
            for (int j = 0; j < NJ; j++) b[j] = tile_load_frag<T>(Bt, b_row0 + j * 32 + li, fc);
#pragma unroll
            for (int i = 0; i < MI; i++)
#pragma unroll
                for (int j = 0; j < NJ; j++) mma32(acc[i][j], a[i], b[j]);
        }
    }
}

template <class T> struct MlpGeom { static constexpr int JC = sizeof(T) == 2 ? 128 : 64; };

template <class T, int C> struct MlpFwdSmem {
    static constexpr int JC = MlpGeom<T>::JC;
    static constexpr int ROWB = 128;                                   // bytes per operand row
    static constexpr int KT_C = C / TileGeom<T>::BK, KT_J = JC / TileGeom<T>::BK;
    static constexpr int A_V2 = KT_C * 128 * ROWB;                     // [128 tokens][C]
    static constexpr int B_W1 = KT_C * JC * ROWB;                      // [JC hidden][C]
    static constexpr int A_H = KT_J * 128 * ROWB;                      // [128 tokens][JC]
    static constexpr int B_W2 = KT_J * C * ROWB;                       // [C out][JC]
    static constexpr int OFF_W1 = A_V2, OFF_H = OFF_W1 + B_W1, OFF_W2 = OFF_H + A_H;
    static constexpr int BYTES = OFF_W2 + B_W2;
    static_assert(64 * (C + 4) * 4 <= A_H + B_W2, "epilogue staging overlays the (contiguous) H and W2 tiles");
};

template <class T, int C>
__global__ void __launch_bounds__(256)
mlp_fwd_kernel(const T* __restrict__ xmid, T* __restrict__ xout, const float* __restrict__ ln_w, const float* __restrict__ ln_b,
               const T* __restrict__ W1, const float* __restrict__ b1, const T* __restrict__ W2, const float* __restrict__ b2,
               const float* __restrict__ gamma, int M, float eps) {
    typedef MlpFwdSmem<T, C> S;
    constexpr int JC = S::JC, HID = 4 * C;
    constexpr int G = C / 8;                       // lanes per row (8 or 16: power of two)
    constexpr int NFX = 128 * G / 256;             // tile frags per thread
    constexpr int NJ1 = JC / 64;                   // fc1 MFMA column blocks per wave (wave tile 64 x JC/2)
    constexpr int NJ2 = C / 64;                    // fc2 MFMA column blocks per wave (wave tile 64 x C/2)
    __shared__ __attribute__((aligned(16))) char smem[S::BYTES];
    char* const Av2 = smem;
    char* const Bw1 = smem + S::OFF_W1;
    char* const Ah = smem + S::OFF_H;
    char* const Bw2 = smem + S::OFF_W2;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, li = lane & 31;
    const int n_tiles = (M + 127) / 128;

    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int m0 = tile * 128;
        // ---- 1. load + LayerNorm (reference maxvit.py:241; biased variance, eps inside the sqrt) ----
        frag_t<T> raw[NFX];
#pragma unroll
        for (int i = 0; i < NFX; i++) {
            const int f = tid + i * 256, row = f / G, cl = f % G;
            const bool ok = m0 + row < M;
            float v[8];
            raw[i] = frag_load<T>(xmid + (size_t)(ok ? m0 + row : 0) * C + cl * 8);
            frag_to_float<T>(raw[i], v);
            float s = 0.f;
#pragma unroll
            for (int e = 0; e < 8; e++) s += v[e];
            const float mean = group_sum(s, G) / (float)C;
            float q = 0.f;
#pragma unroll
            for (int e = 0; e < 8; e++) { const float d = v[e] - mean; q += d * d; }
            const float rstd = 1.0f / sqrtf(group_sum(q, G) / (float)C + eps);
            float o[8];
#pragma unroll
            for (int e = 0; e < 8; e++) o[e] = ok ? (v[e] - mean) * rstd * ln_w[cl * 8 + e] + ln_b[cl * 8 + e] : 0.f;
            tile_store_frag<T>(opm_subtile<T>(Av2, 128, cl / TileGeom<T>::FPR), row, cl % TileGeom<T>::FPR, frag_from_float<T>(o));
        }

        f32x16 acc2[2][NJ2];
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
            for (int j = 0; j < NJ2; j++) acc_zero(acc2[i][j]);

        // ---- 2. hidden chunks ----
        for (int j0 = 0; j0 < HID; j0 += JC) {
            opm_stage<T>(Bw1, W1 + (size_t)j0 * C, C, JC, C, tid);              // W1 rows j0..j0+JC-1, all C columns
            opm_stage<T>(Bw2, W2 + j0, HID, C, JC, tid);                        // W2[:, j0..j0+JC-1]
            __syncthreads();                                                     // v2, W1_j, W2_j visible; previous H consumed
            f32x16 acc1[2][NJ1];
#pragma unroll
            for (int i = 0; i < 2; i++)
#pragma unroll
                for (int j = 0; j < NJ1; j++) acc_zero(acc1[i][j]);
            opm_mma<T, 2, NJ1>(acc1, Av2, 128, wm * 64, Bw1, JC, wn * (JC / 2), C, lane);
            // GELU epilogue in registers -> H as the A operand of fc2
#pragma unroll
            for (int jb = 0; jb < NJ1; jb++) {
                const int n = wn * (JC / 2) + jb * 32 + li;
                const float bias = b1[j0 + n];
#pragma unroll
                for (int i = 0; i < 2; i++)
#pragma unroll
                    for (int r = 0; r < 16; r++)
                        *opm_elem<T>(Ah, 128, wm * 64 + i * 32 + acc_row(r, lane), n) = (T)gelu_f(acc1[i][jb][r] + bias);
            }
            __syncthreads();
            opm_mma<T, 2, NJ2>(acc2, Ah, 128, wm * 64, Bw2, C, wn * (C / 2), JC, lane);
            __syncthreads();                                                     // H / W tiles free for the next chunk
        }

        // ---- 3. epilogue: LayerScale + residual (maxvit.py:51-53,269), 64 tile rows per staging pass ----
        float* stage = reinterpret_cast<float*>(Ah);
        constexpr int LDS_LD = C + 4;
#pragma unroll
        for (int i = 0; i < 2; i++) {
            if (i) __syncthreads();
#pragma unroll
            for (int jb = 0; jb < NJ2; jb++)
#pragma unroll
                for (int r = 0; r < 16; r++)
                    stage[(wm * 32 + acc_row(r, lane)) * LDS_LD + wn * (C / 2) + jb * 32 + li] = acc2[i][jb][r];
            __syncthreads();
            // thread (row, chunk) mapping of step 1: frag slot q of this thread is tile row (tid + 256 q)/G — the rows of
            // pass i are those with ((row>>5)&1) == i, i.e. every thread owns NFX/2 of them
#pragma unroll
            for (int q = 0; q < NFX; q++) {
                const int f = tid + q * 256, row = f / G, cl = f % G;
                if (((row >> 5) & 1) != i) continue;
                const int srow = (row >> 6) * 32 + (row & 31);
                if (m0 + row < M) {
                    float v[8], res[8];
                    frag_to_float<T>(raw[q], res);
#pragma unroll
                    for (int h = 0; h < 2; h++) {
                        const f32x4 t = *reinterpret_cast<const f32x4*>(stage + srow * LDS_LD + cl * 8 + h * 4);
                        v[h * 4 + 0] = t[0]; v[h * 4 + 1] = t[1]; v[h * 4 + 2] = t[2]; v[h * 4 + 3] = t[3];
                    }
#pragma unroll
                    for (int e = 0; e < 8; e++) v[e] = res[e] + gamma[cl * 8 + e] * (v[e] + b2[cl * 8 + e]);
                    frag_store<T>(xout + (size_t)(m0 + row) * C + cl * 8, frag_from_float<T>(v));
                }
            }
        }
        __syncthreads();            // staging (Ah) and Av2 are rewritten by the next tile
    }
}

}  // namespace rvt
