// Parameter-side kernels: everything that turns the fp32 master parameters of the nn.Module (reference names and
// shapes) into the layouts the compute kernels read, and the raw weight-gradient products back into
// gradients of those parameters.  Both directions are TABLE driven: the host builds an array of descriptors once (the
// pointers are persistent: parameter storage, packed-weight buffers, gradient buckets) and one launch walks all of
// them, so an optimisation step costs one "pack" launch and one "finalize" launch per stage instead of a few hundred
// framework micro-kernels.
#pragma once
#include "common.hpp"

namespace rvt {

// element-wise gather from one fp32 source into one destination
enum PackKind {
    PACK_COPY = 0,           // dst[i] = src[i]
    PACK_TRANSPOSE = 1,      // src [R][K] -> dst [K][R], optionally * scale[r]      (d0 = R, d1 = K)
    PACK_CONV_FWD = 2,       // src [Cout][Cin][k][k] -> dst [Cout][k*k*cp] tap-major, cin fastest, zero padded (d0..d3 = Cout,Cin,k,cp)
    PACK_CONV_DGRAD = 3,     // src [Cout][Cin][k][k] -> dst [Cin][nky*nkx*Cout] of one stride-parity class (d0..d4 = Cout,Cin,k,nky,nkx)
    PACK_LSTM_ROWS = 4,      // src [4C][K] -> dst rows interleaved n' = (c/8)*32 + gate*8 + c%8            (d0 = C, d1 = K)
    PACK_CONV_WGRAD_ACC = 5, // src [Cout][k*k*cp] (raw fp32 product) -> dst [Cout][Cin][k][k] += ...      (d0..d3 = Cout,Cin,k,cp)
    PACK_CONV_DGRAD4 = 6     // src [Cout][Cin][3][3] -> dst [(py,px,ci)][(da,db,co)] of the 2 x 2-block input gradient (ppgemm.hpp
                             // GATHER; d0, d1 = Cout, Cin): w[co][ci][ky][kx], ky = 1 | 2, 0 for py = 0 | 1 and da = 0, 1; else 0
};

struct PackDesc {             // 96 bytes; mirrored by rvt_amd/weights.py (numpy structured dtype)
    const float* src;
    void* dst;
    const float* scale;
    long long n;              // destination elements
    int kind;
    int out_f32;              // destination is fp32 whatever the launch dtype
    int d[5];
    int ky[4], kx[4];
    unsigned block0;          // first block of this descriptor within the launch
};
static_assert(sizeof(PackDesc) == 96, "PackDesc layout is part of the C ABI");

// one "logical block" = 1024 consecutive destination elements of one descriptor (block0 counts them): the unit the host tables are
// built in.  A workgroup serves PACK_LBPW consecutive logical blocks: it finds the descriptor of its first one by COUNTING the
// descriptors that start at or before it (every thread looks at its share: one round of loads) and walks on from there - a
// launch over the 18.5 M parameters of RVT-Base is 30 k logical blocks, and one workgroup per logical block spent its time
// on the descriptor look-up and the launch of the workgroup itself (0.21 ms per step; 0.07 ms for the plain copies alone).
constexpr int PACK_LBPW = 8;

template <class T>
__device__ __forceinline__ void pack_logical_block(const PackDesc& d, unsigned lbi, float (*tile)[33]) {
    // (32-bit element indices: a packed weight has far fewer than 2^32 elements, and the index arithmetic below is all divisions)
    const unsigned i0 = (lbi * 256u + threadIdx.x) * 4u, n32 = (unsigned)d.n;
    if (d.kind == PACK_TRANSPOSE && ((d.d[0] | d.d[1]) & 31) == 0) {
        // logical block = one 32 x 32 tile (r fastest), through LDS: 128-byte runs on the source side AND on the destination side
        // (every weight of the backbone takes this path; the two forms below serve odd shapes).  d is uniform: every thread is here.
        const unsigned R = (unsigned)d.d[0], K = (unsigned)d.d[1], ntr = R >> 5;
        const unsigned tk = lbi / ntr, tr = lbi - tk * ntr, c = threadIdx.x & 31, r0 = threadIdx.x >> 5;
        __syncthreads();                                   // (the previous logical block's reads of the tile)
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const unsigned row = 32 * tr + r0 + 8 * j;
            float v = d.src[(size_t)row * K + 32 * tk + c];
            if (d.scale) v *= d.scale[row];
            tile[r0 + 8 * j][c] = v;
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const unsigned krow = r0 + 8 * j;
            const size_t o = (size_t)(32 * tk + krow) * R + 32 * tr + c;
            const float v = tile[c][krow];
            if (d.out_f32) reinterpret_cast<float*>(d.dst)[o] = v;
            else reinterpret_cast<T*>(d.dst)[o] = (T)v;
        }
        return;
    }
    if (d.kind == PACK_TRANSPOSE && (d.d[1] & 3) == 0 && ((size_t)d.src & 15) == 0) {
        // a thread takes FOUR CONSECUTIVE k of one source row r (one 16-byte read) and writes them to four destination rows; the
        // threads of a wave walk r, so every one of the four stores is a contiguous run (the element-order form reads the source
        // with stride K: one 4-byte word per fetched line)
        const unsigned R = (unsigned)d.d[0], K = (unsigned)d.d[1], q = i0 >> 2;
        if (i0 >= n32) return;
        const unsigned kq = q / R, r = q - kq * R;
        f32x4 v4 = *reinterpret_cast<const f32x4*>(d.src + (size_t)r * K + 4 * kq);
        if (d.scale) v4 *= d.scale[r];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const size_t o = (size_t)(4 * kq + u) * R + r;
            if (d.out_f32) reinterpret_cast<float*>(d.dst)[o] = v4[u];
            else reinterpret_cast<T*>(d.dst)[o] = (T)v4[u];
        }
        return;
    }
    if (d.kind == PACK_COPY && ((size_t)d.src & 15) == 0 && ((size_t)d.dst & 15) == 0 && i0 + 4 <= n32) {
        const f32x4 v4 = *reinterpret_cast<const f32x4*>(d.src + i0);
        if (d.out_f32) *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(d.dst) + i0) = v4;
        else {
            typedef __attribute__((ext_vector_type(4))) T vec4;
            vec4 o;
#pragma unroll
            for (int u = 0; u < 4; u++) o[u] = (T)v4[u];
            *reinterpret_cast<vec4*>(reinterpret_cast<T*>(d.dst) + i0) = o;
        }
        return;
    }
    if (d.kind == PACK_LSTM_ROWS && (d.d[1] & 3) == 0 && ((size_t)d.src & 15) == 0 && ((size_t)d.dst & 15) == 0 && i0 < n32) {
        // four consecutive k of one (interleaved) row: one 16-byte read, one store
        const unsigned C = (unsigned)d.d[0], K = (unsigned)d.d[1];
        const unsigned np = i0 / K, kc = i0 - np * K;
        const unsigned c = (np >> 5) * 8 + (np & 7), gate = (np & 31) >> 3;
        const f32x4 v4 = *reinterpret_cast<const f32x4*>(d.src + (size_t)(gate * C + c) * K + kc);
        if (d.out_f32) *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(d.dst) + i0) = v4;
        else {
            typedef __attribute__((ext_vector_type(4))) T vec4;
            vec4 o;
#pragma unroll
            for (int u = 0; u < 4; u++) o[u] = (T)v4[u];
            *reinterpret_cast<vec4*>(reinterpret_cast<T*>(d.dst) + i0) = o;
        }
        return;
    }
#pragma unroll
    for (int u = 0; u < 4; u++) {
        const unsigned i = i0 + u;
        if (i >= n32) return;
        float v = 0.f;
        const unsigned o = i;
        switch (d.kind) {
        case PACK_COPY: v = d.src[i]; break;
        case PACK_TRANSPOSE: {
            const int R = d.d[0], K = d.d[1];
            const int k = (int)(i / R), r = (int)(i % R);
            v = d.src[(size_t)r * K + k];
            if (d.scale) v *= d.scale[r];
            break;
        }
        case PACK_CONV_FWD: {
            const int Cin = d.d[1], k = d.d[2], cp = d.d[3];
            const int kk = k * k * cp;
            const int co = (int)(i / kk), rem = (int)(i % kk);
            const int tap = rem / cp, ci = rem % cp;
            const int ky = tap / k, kx = tap % k;
            v = ci < Cin ? d.src[(((size_t)co * Cin + ci) * k + ky) * k + kx] : 0.f;
            break;
        }
        case PACK_CONV_DGRAD: {
            const int Cout = d.d[0], Cin = d.d[1], k = d.d[2], nkx = d.d[4];
            const int per = d.d[3] * nkx * Cout;
            const int ci = (int)(i / per), rem = (int)(i % per);
            const int ab = rem / Cout, co = rem % Cout;
            const int a = ab / nkx, b = ab % nkx;
            v = d.src[(((size_t)co * Cin + ci) * k + d.ky[a & 3]) * k + d.kx[b & 3]];
            break;
        }
        case PACK_LSTM_ROWS: {
            const int C = d.d[0], K = d.d[1];
            const int np = (int)(i / K), kc = (int)(i % K);
            const int c = (np / 32) * 8 + np % 8, gate = (np % 32) / 8;
            v = d.src[(size_t)(gate * C + c) * K + kc];
            break;
        }
        case PACK_CONV_DGRAD4: {
            const int Cout = d.d[0], Cin = d.d[1];
            const int n = (int)(i / (4 * Cout)), kc = (int)(i % (4 * Cout));
            const int cls = n / Cin, ci = n % Cin, tap = kc / Cout, co = kc % Cout;
            const int py = cls >> 1, px = cls & 1, da = tap >> 1, db = tap & 1;
            const bool on = (py == 1 || da == 0) && (px == 1 || db == 0);
            const int ky = py == 0 ? 1 : (da == 0 ? 2 : 0), kx = px == 0 ? 1 : (db == 0 ? 2 : 0);
            v = on ? d.src[(((size_t)co * Cin + ci) * 3 + ky) * 3 + kx] : 0.f;
            break;
        }
        case PACK_CONV_WGRAD_ACC: {
            const int Cin = d.d[1], k = d.d[2], cp = d.d[3];
            const int per = Cin * k * k;
            const int co = (int)(i / per), rem = (int)(i % per);
            const int ci = rem / (k * k), tap = rem % (k * k);
            v = d.src[(size_t)co * (k * k * cp) + (size_t)tap * cp + ci];
            float* o32 = reinterpret_cast<float*>(d.dst);
            o32[o] += v;
            continue;
        }
        default: break;
        }
        if (d.out_f32) reinterpret_cast<float*>(d.dst)[o] = v;
        else reinterpret_cast<T*>(d.dst)[o] = (T)v;
    }
}

template <class T>
__global__ void __launch_bounds__(256)
pack_table_kernel(const PackDesc* __restrict__ descs, int nd, unsigned total_blocks) {
    const unsigned lb0 = blockIdx.x * (unsigned)PACK_LBPW;
    const unsigned lb1 = lb0 + PACK_LBPW < total_blocks ? lb0 + PACK_LBPW : total_blocks;
    __shared__ int s_cnt;
    __shared__ float tile[32][33];
    if (threadIdx.x == 0) s_cnt = 0;
    __syncthreads();
    int cnt = 0;
    for (int j = threadIdx.x; j < nd; j += 256) cnt += descs[j].block0 <= lb0 ? 1 : 0;       // (block0 is ascending, descs[0].block0 = 0)
    if (cnt) atomicAdd(&s_cnt, cnt);
    __syncthreads();
    int lo = wave_uniform(s_cnt - 1);
    PackDesc d = descs[lo];
    unsigned next0 = lo + 1 < nd ? descs[lo + 1].block0 : 0xffffffffu;
    for (unsigned lb = lb0; lb < lb1; lb++) {
        while (lb >= next0) {
            lo++;
            d = descs[lo];
            next0 = lo + 1 < nd ? descs[lo + 1].block0 : 0xffffffffu;
        }
        pack_logical_block<T>(d, lb - d.block0, tile);
    }
}

// LayerScale (reference maxvit.py:51-53) folded out of the weight gradients: the branch is y = gamma * (a W^T + b) and the
// weight-gradient GEMMs deliver the raw products S[c][k] = sum_tok dy[tok][c] a[tok][k] and cs[c] = sum_tok dy[tok][c] of
// the *outer* cotangent dy (the dgrad weights carry gamma instead).  Then
//   dW[c][k] += gamma[c] S[c][k],   db[c] += gamma[c] cs[c],   dgamma[c] += sum_k W[c][k] S[c][k] + b[c] cs[c].
struct LayerScaleDesc {       // 80 bytes
    const float* S; const float* cs; const float* W; const float* b; const float* gamma;
    float* dW; float* db; float* dgamma;
    int C, K;
    unsigned block0; int pad;
};
static_assert(sizeof(LayerScaleDesc) == 80, "LayerScaleDesc layout is part of the C ABI");

__global__ void __launch_bounds__(256)
layerscale_grad_kernel(const LayerScaleDesc* __restrict__ descs, int nd) {
    int lo = 0, hi = nd - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (descs[mid].block0 <= blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const LayerScaleDesc d = descs[lo];
    const int c = (int)(blockIdx.x - d.block0);
    if (c >= d.C) return;
    const float g = d.gamma[c];
    float acc = 0.f;
    for (int k = threadIdx.x; k < d.K; k += 256) {
        const float s = d.S[(size_t)c * d.K + k];
        d.dW[(size_t)c * d.K + k] += g * s;
        acc += d.W[(size_t)c * d.K + k] * s;
    }
    __shared__ float red[256];
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int m = 128; m > 0; m >>= 1) {
        if ((int)threadIdx.x < m) red[threadIdx.x] += red[threadIdx.x + m];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const float cs = d.cs[c];
        d.dgamma[c] += red[0] + d.b[c] * cs;
        d.db[c] += g * cs;
    }
}

}  // namespace rvt
