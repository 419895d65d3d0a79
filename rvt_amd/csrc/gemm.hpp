// One MFMA GEMM engine for every contraction on the RVT hot path.
//
//   D[m][n] (+)= sum_k A[m][k] * B[n][k]          (both operands addressed "row, k")
//
// Operands are *sources*: objects that map (row, column-segment) to a pointer to 8+ contiguous
// elements, or nullptr for padding.  That one abstraction covers
//   PlainSrc   row-major matrices (linear layers, weights),
//   ConcatSrc  [x_t | h_{t-1}] of the ConvLSTM 1x1 conv (reference rnn.py:52,55) without a cat copy,
//   Im2colSrc  the strided overlapping conv of the down-sampling stem (reference maxvit.py:160-168),
//   DgradSrc   its input-gradient gather for one stride-parity class of input pixels.
// Two loaders stage a 128-row x 128-byte tile into swizzled LDS:
//   NT: the contraction index is the source's column (contiguous) -> 16/32-byte vector loads;
//   TN: the contraction index is the source's ROW (token)  -> 4-byte loads from 8 consecutive
//       tokens, transposed in registers (weight gradients: dW = dY^T X, contraction over tokens).
// Results leave through an LDS-staged epilogue that hands 8 (or 32) consecutive columns of one row
// to an epilogue functor (bias / GELU' / LayerScale+residual / LSTM gates / fp32 atomics for split-K).
//
// Tile: 128 x BN (BN = 64 or 128) per 256-thread workgroup, 4 waves as 2x2, each wave
// (64 x BN/2) = 2 x (BN/64) MFMA 32x32 blocks; K tile = 128 bytes (64 bf16 / 32 f32), register-staged
// double buffering (global->VGPR for tile k+1 is issued before the MFMAs of tile k).
#pragma once
#include <stdlib.h>
#include <type_traits>
#include "common.hpp"

#ifndef TN_WAVES
#define TN_WAVES 2      // weight-gradient (TN) kernels: registers capped for two workgroups per CU
#endif
namespace rvt {

// ------------------------------------------------------------------------------------------------
// sources
// ------------------------------------------------------------------------------------------------
template <class T> struct PlainSrc {
    const T* p; int ld; int rows; int cols;
    static constexpr bool LINEAR = true;               // element (row, seg, off) = linear_base(seg)[row * linear_ld() + off]
    static constexpr bool UNIT_LINEAR = false;
    static constexpr bool SPATIAL_REUSE = false;
    __device__ __forceinline__ bool unit_linear(int, int, const T*&, size_t&) const { return false; }
    __device__ __forceinline__ const T* linear_base(int) const { return p; }
    __device__ __forceinline__ int linear_ld() const { return ld; }
    typedef const T* Ctx;
    __device__ __forceinline__ Ctx row_ctx(int m) const { return (m >= 0 && m < rows) ? p + (size_t)m * ld : nullptr; }
    __device__ __forceinline__ void split(int kcol, int& seg, int& off) const { seg = 0; off = kcol; }
    __device__ __forceinline__ const T* seg_ptr(const Ctx& c, int) const { return c; }
    __device__ __forceinline__ const T* safe() const { return p; }
    // context of row m+1 given the (valid) context of row m
    __device__ __forceinline__ Ctx advance(const Ctx& c) const { return c + ld; }
};

template <class T> struct ConcatSrc {   // [x | h], both [rows][C]
    const T* x; const T* h; int C; int rows; int cols;  // cols = 2C
    static constexpr bool LINEAR = true;
    static constexpr bool UNIT_LINEAR = false;
    static constexpr bool SPATIAL_REUSE = false;
    __device__ __forceinline__ bool unit_linear(int, int, const T*&, size_t&) const { return false; }
    __device__ __forceinline__ const T* linear_base(int seg) const { return seg ? h : x; }
    __device__ __forceinline__ int linear_ld() const { return C; }
    typedef int Ctx;
    __device__ __forceinline__ Ctx row_ctx(int m) const { return (m >= 0 && m < rows) ? m : -1; }
    __device__ __forceinline__ void split(int kcol, int& seg, int& off) const { seg = kcol >= C; off = kcol - seg * C; }
    __device__ __forceinline__ const T* seg_ptr(const Ctx& c, int seg) const {
        const T* q = (seg ? h : x) + (size_t)(c < 0 ? 0 : c) * C;
        return c < 0 ? nullptr : q;
    }
    __device__ __forceinline__ const T* safe() const { return x; }
    __device__ __forceinline__ Ctx advance(const Ctx& c) const { return c + 1; }
};

// rows = output pixels (frame, oy, ox); columns = (ky, kx, cin) with cin fastest; input is [F][H][W][Cin]
template <class T> struct Im2colSrc {
    const T* p; int H, W, Cin, Ho, Wo, kw, stride, pad; int rows; int cols;  // cols = kh*kw*Cin
    FastDiv dHoWo, dWo, dkw, dCin;
    static constexpr bool LINEAR = false;
    static constexpr bool SPATIAL_REUSE = true;
    __device__ __forceinline__ const T* linear_base(int) const { return p; }
    __device__ __forceinline__ int linear_ld() const { return 0; }
    struct Ctx { int base; int iy0; int ix0; };
    __device__ __forceinline__ Ctx row_ctx(int m) const {
        Ctx c;
        const bool ok = (m >= 0) & (m < rows);
        uint32_t f, rem, oy, ox;
        dHoWo.divmod((uint32_t)(ok ? m : 0), f, rem);
        dWo.divmod(rem, oy, ox);
        c.base = ok ? (int)f * H * W : -1; c.iy0 = (int)oy * stride - pad; c.ix0 = (int)ox * stride - pad;
        return c;
    }
    __device__ __forceinline__ void split(int kcol, int& seg, int& off) const {
        uint32_t q, r; dCin.divmod((uint32_t)kcol, q, r); seg = (int)q; off = (int)r;
    }
    __device__ __forceinline__ const T* seg_ptr(const Ctx& c, int seg) const {
        uint32_t ky, kx; dkw.divmod((uint32_t)seg, ky, kx);
        const int iy = c.iy0 + (int)ky, ix = c.ix0 + (int)kx;
        const bool ok = (c.base >= 0) & (iy >= 0) & (iy < H) & (ix >= 0) & (ix < W);
        const size_t pix = ok ? (size_t)c.base + (size_t)iy * W + ix : 0;      // branch-free: clamp, then select
        const T* q = p + pix * Cin;
        return ok ? q : nullptr;
    }
    __device__ __forceinline__ const T* safe() const { return p; }
    // Eight consecutive output pixels tok0..tok0+7 of tap `seg`: when they share an output row and none of them falls
    // into the padding, their source pixels are `stride` apart on one input row — one address and a constant step
    // instead of eight decode / bounds-check / wrap sequences (the transposing loader's unit, see TNLoader::load).
    static constexpr bool UNIT_LINEAR = true;
    __device__ __forceinline__ bool unit_linear(int tok0, int seg, const T*& p0, size_t& step) const {
        if ((Wo & 7) != 0 || tok0 + 8 > rows) return false;           // units never straddle rows iff Wo % 8 == 0
        uint32_t f, rem, oy, ox, ky, kx;
        dHoWo.divmod((uint32_t)tok0, f, rem);
        dWo.divmod(rem, oy, ox);
        dkw.divmod((uint32_t)seg, ky, kx);
        const int iy = (int)oy * stride - pad + (int)ky;
        const int ix = (int)ox * stride - pad + (int)kx;
        if (iy < 0 || iy >= H || ix < 0 || ix + 7 * stride >= W) return false;
        p0 = p + ((size_t)f * H * W + (size_t)iy * W + ix) * Cin;
        step = (size_t)stride * Cin;
        return true;
    }
    // next output pixel in raster order: x+1, wrapping to the next row / frame (no divisions)
    __device__ __forceinline__ Ctx advance(const Ctx& c) const {
        Ctx n = c;
        n.ix0 += stride;
        const bool wrap_x = n.ix0 >= Wo * stride - pad;
        n.ix0 = wrap_x ? -pad : n.ix0;
        n.iy0 += wrap_x ? stride : 0;
        const bool wrap_y = n.iy0 >= Ho * stride - pad;
        n.iy0 = wrap_y ? -pad : n.iy0;
        n.base += wrap_y ? H * W : 0;
        return n;
    }
};

// Input-gradient gather of a strided conv for ONE parity class (py,px) of input pixels
// y = s*yy+py, x = s*xx+px.  Only the taps ky ≡ (py+pad) mod s contribute:  oy = (y+pad-ky)/s.
// rows = (frame, yy, xx) over the class; columns = (a, b, cout) over the class's taps Ky[a], Kx[b].
template <class T> struct DgradSrc {
    const T* dy; int Ho, Wo, Cout; int Hc, Wc;     // class extent
    int s, pad, py, px; int nky, nkx; int ky[4], kx[4];
    int rows; int cols;                             // cols = nky*nkx*Cout
    FastDiv dHcWc, dWc, dCout;
    static constexpr bool LINEAR = false;
    static constexpr bool UNIT_LINEAR = false;
    static constexpr bool SPATIAL_REUSE = true;
    __device__ __forceinline__ bool unit_linear(int, int, const T*&, size_t&) const { return false; }
    __device__ __forceinline__ const T* linear_base(int) const { return dy; }
    __device__ __forceinline__ int linear_ld() const { return 0; }
    struct Ctx { int f; int y; int x; };
    __device__ __forceinline__ Ctx row_ctx(int m) const {
        Ctx c;
        const bool ok = (m >= 0) & (m < rows);
        uint32_t f, rem, yy, xx;
        dHcWc.divmod((uint32_t)(ok ? m : 0), f, rem);
        dWc.divmod(rem, yy, xx);
        c.f = ok ? (int)f : -1; c.y = (int)yy * s + py; c.x = (int)xx * s + px;
        return c;
    }
    __device__ __forceinline__ void split(int kcol, int& seg, int& off) const {
        uint32_t q, r; dCout.divmod((uint32_t)kcol, q, r); seg = (int)q; off = (int)r;
    }
    __device__ __forceinline__ const T* seg_ptr(const Ctx& c, int seg) const {
        const int a = nkx == 1 ? seg : (nkx == 2 ? seg >> 1 : seg / nkx), b = seg - a * nkx;
        const int ny = c.y + pad - ky[a & 3], nx = c.x + pad - kx[b & 3];
        const int oy = s == 2 ? ny >> 1 : ny / s, ox = s == 2 ? nx >> 1 : nx / s;
        const bool ok = (c.f >= 0) & (ny >= 0) & (nx >= 0) & (oy < Ho) & (ox < Wo);
        const size_t pix = ok ? ((size_t)c.f * Ho + oy) * Wo + ox : 0;
        const T* q = dy + pix * Cout;
        return ok ? q : nullptr;
    }
    __device__ __forceinline__ const T* safe() const { return dy; }
    __device__ __forceinline__ Ctx advance(const Ctx& c) const {
        Ctx n = c;
        n.x += s;
        const bool wrap_x = n.x >= Wc * s + px;                          // x = s*xx+px with xx < Wc
        n.x = wrap_x ? px : n.x;
        n.y += wrap_x ? s : 0;
        const bool wrap_y = n.y >= Hc * s + py;
        n.y = wrap_y ? py : n.y;
        n.f += wrap_y ? 1 : 0;
        return n;
    }
};

// element-wise transforms applied to loaded operand values
struct XfNone { __device__ __forceinline__ float operator()(float v) const { return v; } static constexpr bool identity = true; };
struct XfGelu { __device__ __forceinline__ float operator()(float v) const { return gelu_f(v); } static constexpr bool identity = false; };

template <class T, class Xf> __device__ __forceinline__ frag_t<T> xf_apply(const frag_t<T>& f, const Xf& xf) {
    if (Xf::identity) return f;
    frag_t<T> o;
#pragma unroll
    for (int i = 0; i < 8; i++) o[i] = (T)xf((float)f[i]);
    return o;
}

// ------------------------------------------------------------------------------------------------
// tile loaders (global -> registers -> swizzled LDS)
// ------------------------------------------------------------------------------------------------
template <class T, int ROWS, class Src, class Xf> struct NTLoader {
    static constexpr int FPR = TileGeom<T>::FPR;
    static constexpr int NF = ROWS * FPR / 256;
    typename Src::Ctx ctx[NF];
    frag_t<T> r[NF];
    bool valid[NF];
    __device__ __forceinline__ void init(const Src& s, int row0, int tid) {
#pragma unroll
        for (int i = 0; i < NF; i++) ctx[i] = s.row_ctx(row0 + (tid + i * 256) / FPR);
    }
    __device__ __forceinline__ void load(const Src& s, const Xf& xf, int k0, int kend, int tid) {
#pragma unroll
        for (int i = 0; i < NF; i++) {
            int fc = (tid + i * 256) % FPR;
            int kcol = k0 + fc * 8;
            // Branch-free: always issue the load (from a clamped, always-valid address) and remember whether the
            // element is real; padding is zeroed by a select in store().  Nothing consumes the loaded registers
            // until store(), so the loads stay in flight across the MFMA phase of the current tile.
            int seg, off;
            s.split(kcol < kend ? kcol : 0, seg, off);
            const T* p = s.seg_ptr(ctx[i], seg);
            const bool ok = (kcol < kend) & (p != nullptr);
            r[i] = frag_load<T>(ok ? p + off : s.safe());
            valid[i] = ok;
        }
    }
    __device__ __forceinline__ void store(char* tile, const Xf& xf, int tid) {
#pragma unroll
        for (int i = 0; i < NF; i++) {
            int u = tid + i * 256;
            frag_t<T> v = r[i];
            if (!Xf::identity) v = xf_apply<T>(v, xf);
            const frag_t<T> z = frag_zero<T>();
            v = valid[i] ? v : z;
            tile_store_frag<T>(tile, u / FPR, u % FPR, v);
        }
    }
};

// Transposing loader: tile row = source COLUMN (feature), contraction = source ROW (token).
// Work unit = 8 features x 8 tokens: eight 16-byte (bf16) / 32-byte (f32) row-contiguous loads, an 8x8
// transpose in registers, eight frag stores (one per feature, 8 consecutive tokens each).  Lanes run along the
// feature axis, so a wave's loads cover whole 256-byte row segments.  A tile has (ROWS/8)*(BK/8) <= 128 units;
// the A loader uses the low threads and the B loader the high threads so both halves of the workgroup load.
template <class T> struct Transpose8;
template <> struct Transpose8<float> {
    static __device__ __forceinline__ void run(const f32x8 (&in)[8], f32x8 (&out)[8]) {
#pragma unroll
        for (int f = 0; f < 8; f++)
#pragma unroll
            for (int j = 0; j < 8; j++) out[f][j] = in[j][f];
    }
};
template <> struct Transpose8<bf16> {
    static __device__ __forceinline__ void run(const bf16x8 (&in)[8], bf16x8 (&out)[8]) {
        // 16-bit 8x8 transpose on packed dwords: out[f].dword[q] = { in[2q].half[f], in[2q+1].half[f] } — one byte
        // permute (v_perm_b32) per output dword
        u32x4 w[8];
#pragma unroll
        for (int j = 0; j < 8; j++) w[j] = *reinterpret_cast<const u32x4*>(&in[j]);
#pragma unroll
        for (int f = 0; f < 8; f++) {
            u32x4 o;
#pragma unroll
            for (int q = 0; q < 4; q++) {
                uint32_t a = w[2 * q][f >> 1], b = w[2 * q + 1][f >> 1];
#ifdef RVT_EMU
                o[q] = (f & 1) ? ((a >> 16) | (b & 0xffff0000u)) : ((a & 0xffffu) | (b << 16));
#else
                o[q] = __builtin_amdgcn_perm(b, a, (f & 1) ? 0x07060302u : 0x05040100u);   // bytes 0-3 = a, 4-7 = b
#endif
            }
            out[f] = *reinterpret_cast<const bf16x8*>(&o);
        }
    }
};

// The eight raw source rows of one unit while they are in flight.  A thread serves the A operand or the B operand,
// never both (wave-uniform), so the two loaders of a kernel share one TNRegs; the K loop keeps TWO of them in flight.
template <class T> struct TNRegs {
    frag_t<T> r[8];
    unsigned vmask;
};

template <class T, int ROWS, class Src, class Xf, bool HIGH> struct TNLoader {
    static constexpr int FPR = TileGeom<T>::FPR;          // token chunks (of 8) per K tile
    static constexpr int FC = ROWS / 8;                   // feature chunks per tile
    static constexpr int NUNITS = FC * FPR;               // <= 128
    static_assert(NUNITS <= 256, "tile too large for one unit per thread");
    typedef TNRegs<T> Regs;
    int u;                                                // this thread's unit or -1
    int seg, off;
    bool fvalid;
    const T* colbase;                                     // LINEAR sources: address of (token 0, this unit's first feature)
    int ldl;
    int soff[8 * TileGeom<T>::CPF];                       // LDS byte offsets of the unit's 8 feature rows (fixed for the launch)
    __device__ __forceinline__ void init(const Src& s, int row0, int tid) {
        u = HIGH ? tid - (256 - NUNITS) : tid;
        if (u >= NUNITS) u = -1;
        seg = 0; off = 0; fvalid = false; colbase = s.safe(); ldl = 0;
        if (u >= 0) {
            int feat = row0 + (u % FC) * 8;
            fvalid = feat < s.cols;
            if (fvalid) s.split(feat, seg, off);
            if (Src::LINEAR) { colbase = s.linear_base(seg) + off; ldl = s.linear_ld(); }
#pragma unroll
            for (int f = 0; f < 8; f++)
#pragma unroll
                for (int c = 0; c < TileGeom<T>::CPF; c++)
                    soff[f * TileGeom<T>::CPF + c] = lds_chunk_off((u % FC) * 8 + f, (u / FC) * TileGeom<T>::CPF + c);
        }
    }
    __device__ __forceinline__ void load(const Src& s, const Xf& xf, int k0, int kend, int tid, Regs& R) {
        if (u < 0) return;
        const int tok0 = k0 + (u / FC) * 8;
        if (!fvalid) { R.vmask = 0u; return; }               // feature padding of a partial tile: zeros, no traffic
        if (Src::LINEAR && tok0 + 8 <= kend) {
            // interior unit of a row-major source (every K tile but a slice's last): eight unconditional loads,
            // one address computation
            const T* p = colbase + (size_t)tok0 * ldl;
#pragma unroll
            for (int j = 0; j < 8; j++) R.r[j] = frag_load<T>(p + (size_t)j * ldl);
            R.vmask = 0xffu;
            return;
        }
        if (Src::UNIT_LINEAR && tok0 + 8 <= kend) {
            const T* p0; size_t step;
            if (s.unit_linear(tok0, seg, p0, step)) {            // interior unit of an im2col source
#pragma unroll
                for (int j = 0; j < 8; j++) R.r[j] = frag_load<T>(p0 + off + (size_t)j * step);
                R.vmask = 0xffu;
                return;
            }
        }
        unsigned vmask = 0;
        // tokens tok0..tok0+7 are consecutive source rows: decode the first, step the rest (validity is by index,
        // so stepping past the last real row is harmless: those loads go to the clamped address)
        typename Src::Ctx c = s.row_ctx(tok0 < kend ? tok0 : 0);
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const int tok = tok0 + j;
            const T* p = s.seg_ptr(c, seg);
            const bool ok = fvalid & (tok < kend) & (p != nullptr);
            R.r[j] = frag_load<T>(ok ? p + off : s.safe());   // branch-free; raw rows, post-processing in store()
            vmask |= ok ? (1u << j) : 0u;
            c = s.advance(c);
        }
        R.vmask = vmask;
    }
    __device__ __forceinline__ void store(char* tile, const Xf& xf, int tid, const Regs& R) {
        if (u < 0) return;
        frag_t<T> in[8], out[8];
#pragma unroll
        for (int j = 0; j < 8; j++) {
            in[j] = R.r[j];
            if (!Xf::identity) in[j] = xf_apply<T>(in[j], xf);
        }
        if (R.vmask != 0xffu) {                               // padding rows / columns -> exact zeros
            const frag_t<T> z = frag_zero<T>();
#pragma unroll
            for (int j = 0; j < 8; j++) in[j] = ((R.vmask >> j) & 1u) ? in[j] : z;
        }
        Transpose8<T>::run(in, out);
#pragma unroll
        for (int f = 0; f < 8; f++) {
            const u32x4* src = reinterpret_cast<const u32x4*>(&out[f]);
#pragma unroll
            for (int c = 0; c < TileGeom<T>::CPF; c++)
                *reinterpret_cast<u32x4*>(tile + soff[f * TileGeom<T>::CPF + c]) = src[c];
        }
    }
};

// ------------------------------------------------------------------------------------------------
// epilogues.  A thread owns ONE column unit (UNIT consecutive columns n .. n+UNIT-1) for the whole tile and a few rows
// of it per 64-row pass, so an epilogue is split into three steps that the kernel schedules separately:
//   cols(n)            per-column parameters (bias, LayerScale gamma) -> registers, once per tile, before the MFMA loop
//   fetch(m, n)        per-element side inputs (residual, GELU', c_prev ...) -> registers; issued for ALL units of the
//                      tile before the next tile's operand prefetch, so that nothing in the store path ever waits on a
//                      load: on gfx950 loads and stores share the in-order vmcnt counter, and a load issued after a
//                      store cannot be waited for without also waiting for that store's acknowledgement from memory
//   apply(m, n, v, aux, cols)   arithmetic + stores
// ------------------------------------------------------------------------------------------------
struct EpNone {};

template <int N> struct EpColVec {
    float v[N];
};
template <int N> __device__ __forceinline__ EpColVec<N> ep_load_cols(const float* p, int n, bool ok) {
    EpColVec<N> c;
    if (p != nullptr && ok) {
#pragma unroll
        for (int q = 0; q < N / 4; q++) {
            f32x4 t = *reinterpret_cast<const f32x4*>(p + n + q * 4);
            c.v[q * 4 + 0] = t[0]; c.v[q * 4 + 1] = t[1]; c.v[q * 4 + 2] = t[2]; c.v[q * 4 + 3] = t[3];
        }
    } else {
#pragma unroll
        for (int i = 0; i < N; i++) c.v[i] = 0.f;
    }
    return c;
}
template <class T> struct EpFragAux {
    frag_t<T> f;
    __device__ __forceinline__ EpFragAux() : f(frag_zero<T>()) {}
};

template <class T> struct EpStore {
    __device__ __forceinline__ void begin_block(int) {}            // out = v (+bias) (+add)
    static constexpr int UNIT = 8;
    typedef EpColVec<8> Cols;
    typedef EpFragAux<T> Aux;
    T* out; int ld; const float* bias; const T* add;
    __device__ __forceinline__ Cols cols(int n, bool ok) const { return ep_load_cols<8>(bias, n, ok); }
    __device__ __forceinline__ Aux fetch(int m, int n) const {
        Aux a;
        if (add) a.f = frag_load<T>(add + (size_t)m * ld + n);
        return a;
    }
    __device__ __forceinline__ void apply(int m, int n, float (&v)[8], const Aux& aux, const Cols& c) const {
        float a[8]; frag_to_float<T>(aux.f, a);
#pragma unroll
        for (int i = 0; i < 8; i++) v[i] += c.v[i] + a[i];
        frag_store<T>(out + (size_t)m * ld + n, frag_from_float<T>(v));
    }
};

// out = act(v * scale[n] + shift[n]): inference-mode BatchNorm folded behind a convolution (scale = gamma * rstd of the running
// statistics, shift = beta - running_mean * scale) + SiLU, applied to the fp32 accumulator (YOLOX BaseConv, network_blocks.py:29-53)
template <class T> struct EpAffineAct {
    __device__ __forceinline__ void begin_block(int) {}
    static constexpr int UNIT = 8;
    struct Cols { EpColVec<8> sc, sh; };
    typedef EpNone Aux;
    T* out; int ld; const float* scale; const float* shift; int act;       // act: 0 none, 1 SiLU
    __device__ __forceinline__ Cols cols(int n, bool ok) const {
        Cols c; c.sc = ep_load_cols<8>(scale, n, ok); c.sh = ep_load_cols<8>(shift, n, ok); return c;
    }
    __device__ __forceinline__ Aux fetch(int, int) const { return Aux(); }
    __device__ __forceinline__ void apply(int m, int n, float (&v)[8], const Aux&, const Cols& c) const {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const float z = fmaf(v[i], c.sc.v[i], c.sh.v[i]);
            v[i] = act == 1 ? z * sigmoid_f(z) : z;
        }
        frag_store<T>(out + (size_t)m * ld + n, frag_from_float<T>(v));
    }
};

template <class T> struct EpScaleRes {
    __device__ __forceinline__ void begin_block(int) {}         // out = res + gamma * (v + bias)     (LayerScale + residual)
    static constexpr int UNIT = 8;
    struct Cols { EpColVec<8> b, g; };
    typedef EpFragAux<T> Aux;
    T* out; const T* res; int ld; const float* bias; const float* gamma;
    __device__ __forceinline__ Cols cols(int n, bool ok) const {
        Cols c; c.b = ep_load_cols<8>(bias, n, ok); c.g = ep_load_cols<8>(gamma, n, ok); return c;
    }
    __device__ __forceinline__ Aux fetch(int m, int n) const {
        Aux a; a.f = frag_load<T>(res + (size_t)m * ld + n); return a;
    }
    __device__ __forceinline__ void apply(int m, int n, float (&v)[8], const Aux& aux, const Cols& c) const {
        float a[8]; frag_to_float<T>(aux.f, a);
#pragma unroll
        for (int i = 0; i < 8; i++) v[i] = a[i] + c.g.v[i] * (v[i] + c.b.v[i]);
        frag_store<T>(out + (size_t)m * ld + n, frag_from_float<T>(v));
    }
};

template <class T> struct EpGeluBwd {
    __device__ __forceinline__ void begin_block(int) {}          // out = v * gelu'(pre[m][n])
    static constexpr int UNIT = 8;
    typedef EpNone Cols;
    typedef EpFragAux<T> Aux;
    T* out; const T* pre; int ld;
    __device__ __forceinline__ Cols cols(int, bool) const { return Cols(); }
    __device__ __forceinline__ Aux fetch(int m, int n) const {
        Aux a; a.f = frag_load<T>(pre + (size_t)m * ld + n); return a;
    }
    __device__ __forceinline__ void apply(int m, int n, float (&v)[8], const Aux& aux, const Cols&) const {
        float a[8]; frag_to_float<T>(aux.f, a);
#pragma unroll
        for (int i = 0; i < 8; i++) v[i] *= gelu_grad_f(a[i]);
        frag_store<T>(out + (size_t)m * ld + n, frag_from_float<T>(v));
    }
};

template <class T> struct EpGeluDual {         // g = gelu(v + bias), gp = gelu'(v + bias)  (gp nullable)
    __device__ __forceinline__ void begin_block(int) {}
    static constexpr int UNIT = 8;
    typedef EpColVec<8> Cols;
    typedef EpNone Aux;
    T* g; T* gp; int ld; const float* bias;
    __device__ __forceinline__ Cols cols(int n, bool ok) const { return ep_load_cols<8>(bias, n, ok); }
    __device__ __forceinline__ Aux fetch(int, int) const { return Aux(); }
    __device__ __forceinline__ void apply(int m, int n, float (&v)[8], const Aux&, const Cols& c) const {
        float a[8], b[8];
#pragma unroll
        for (int i = 0; i < 8; i++) v[i] += c.v[i];
        gelu_both_8(v, a, b);
        frag_store<T>(g + (size_t)m * ld + n, frag_from_float<T>(a));
        if (gp) frag_store<T>(gp + (size_t)m * ld + n, frag_from_float<T>(b));
    }
};

template <class T> struct EpMul {              // out = v * mul[m][n]
    __device__ __forceinline__ void begin_block(int) {}
    static constexpr int UNIT = 8;
    typedef EpNone Cols;
    typedef EpFragAux<T> Aux;
    T* out; const T* mul; int ld;
    __device__ __forceinline__ Cols cols(int, bool) const { return Cols(); }
    __device__ __forceinline__ Aux fetch(int m, int n) const {
        Aux a; a.f = frag_load<T>(mul + (size_t)m * ld + n); return a;
    }
    __device__ __forceinline__ void apply(int m, int n, float (&v)[8], const Aux& aux, const Cols&) const {
        float a[8]; frag_to_float<T>(aux.f, a);
#pragma unroll
        for (int i = 0; i < 8; i++) v[i] *= a[i];
        frag_store<T>(out + (size_t)m * ld + n, frag_from_float<T>(v));
    }
};

template <class T> struct EpSplit2 {
    __device__ __forceinline__ void begin_block(int) {}           // columns [0,C) -> out0, [C,2C) -> out1 (both ld = C)
    static constexpr int UNIT = 8;
    typedef EpNone Cols;
    typedef EpNone Aux;
    T* out0; T* out1; int C;
    __device__ __forceinline__ Cols cols(int, bool) const { return Cols(); }
    __device__ __forceinline__ Aux fetch(int, int) const { return Aux(); }
    __device__ __forceinline__ void apply(int m, int n, float (&v)[8], const Aux&, const Cols&) const {
        T* o = n < C ? out0 + (size_t)m * C + n : out1 + (size_t)m * C + (n - C);
        frag_store<T>(o, frag_from_float<T>(v));
    }
};

struct EpAtomicF32 {                           // direct atomic accumulation (kept for tiny problems / no workspace)
    static constexpr int UNIT = 8;
    typedef EpNone Cols;
    typedef EpNone Aux;
    __device__ __forceinline__ void begin_block(int) {}
    float* out; int ld;
    __device__ __forceinline__ Cols cols(int, bool) const { return Cols(); }
    __device__ __forceinline__ Aux fetch(int, int) const { return Aux(); }
    __device__ __forceinline__ void apply(int m, int n, float (&v)[8], const Aux&, const Cols&) const {
#pragma unroll
        for (int i = 0; i < 8; i++) atomicAdd(out + (size_t)m * ld + n + i, v[i]);
    }
};

// two-stage split-K: every K-slice stores its partial tile to ws[slice][M][N] with plain stores; a small
// reduction kernel folds the slices afterwards.  (Device-scope float atomics execute memory-side on the
// 8-XCD MI355X — ~64 B of fabric traffic and ~0.4 ns each chip-wide — so they are kept off the hot path.)
struct EpPartialStore {
    static constexpr int UNIT = 8;
    typedef EpNone Cols;
    typedef EpNone Aux;
    float* ws; int ld; size_t slice_elems; int slice;
    __device__ __forceinline__ void begin_block(int split) { slice = split; }
    __device__ __forceinline__ Cols cols(int, bool) const { return Cols(); }
    __device__ __forceinline__ Aux fetch(int, int) const { return Aux(); }
    __device__ __forceinline__ void apply(int m, int n, float (&v)[8], const Aux&, const Cols&) const {
        float* o = ws + (size_t)slice * slice_elems + (size_t)m * ld + n;
        *reinterpret_cast<f32x4*>(o) = f32x4{v[0], v[1], v[2], v[3]};
        *reinterpret_cast<f32x4*>(o + 4) = f32x4{v[4], v[5], v[6], v[7]};
    }
};

// Split-K folds.  A workgroup owns 32 consecutive outputs x 8 slice lanes: thread (e, g) sums slices g, g + 8, ... of output e
// (four independent chains), the eight partial sums meet in LDS in a fixed order (deterministic).  The one-thread-per-output
// form walked all slices in one thread: nsplit / 8 dependent round trips - 10-20 us for the 256-512 slices of the small
// weight gradients, 1.2 ms per Base step over 73 launches.  Grid: reduce_grid(count) workgroups (any grid is correct).
inline int reduce_grid(size_t count) {
    size_t g = (count + 31) / 32;
    return (int)(g < 1 ? 1 : (g > 8192 ? 8192 : g));
}
template <class Addr>
__device__ __forceinline__ void reduce_rows_32x8(const float* __restrict__ ws, int nrec, size_t count, const Addr& addr,
                                                  float* __restrict__ out, int t_cols, unsigned bid, unsigned nblk) {
    __shared__ float red[8][33];
    const int e = threadIdx.x & 31, g = threadIdx.x >> 5;
    for (size_t i0 = (size_t)bid * 32; i0 < count; i0 += (size_t)nblk * 32) {
        const size_t i = i0 + e;
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        if (i < count) {
            int s = g;
            for (; s + 24 < nrec; s += 32) {
                a0 += ws[addr(s) + i]; a1 += ws[addr(s + 8) + i]; a2 += ws[addr(s + 16) + i]; a3 += ws[addr(s + 24) + i];
            }
            for (; s < nrec; s += 8) a0 += ws[addr(s) + i];
        }
        red[g][e] = (a0 + a1) + (a2 + a3);
        __syncthreads();
        if (g == 0 && i < count) {
            float t = 0.f;
#pragma unroll
            for (int k = 0; k < 8; k++) t += red[k][e];
            size_t o = i;
            if (t_cols > 0) { const size_t r = i / t_cols, c = i % t_cols; o = c * (count / t_cols) + r; }
            out[o] += t;
        }
        __syncthreads();
    }
}
// out[i] += sum_s ws[s][i]; with t_cols > 0 the partial tiles are [count / t_cols][t_cols] and `out` is their transpose
// (static: gemm.hpp is included by several translation units of the gfx950 build)
static __global__ void __launch_bounds__(256)
splitk_reduce_kernel(const float* __restrict__ ws, float* __restrict__ out, int nsplit, size_t count, int t_cols) {
    reduce_rows_32x8(ws, nsplit, count, [count](int s) { return (size_t)s * count; }, out, t_cols, blockIdx.x, gridDim.x);
}
// out[i] += sum_s ws[s * stride + i], i < count  (per-workgroup partial RECORDS of `stride` floats each)
static __global__ void __launch_bounds__(256)
strided_reduce_kernel(const float* __restrict__ ws, float* __restrict__ out, int nrec, size_t stride, size_t count) {
    reduce_rows_32x8(ws, nrec, count, [stride](int s) { return (size_t)s * stride; }, out, 0, blockIdx.x, gridDim.x);
}
// Several folds of ONE producer launch in one launch (round 6): a weight gradient and its bias column sums, the four outputs of
// an MLP weight-gradient kernel, a ConvLSTM weight block and its bias rows.  73 fold launches per RVT-Base step were 0.8 ms, most
// of it the launches themselves (a bias fold is one or two workgroups).  Job j owns the workgroups [block0, block0 + blocks).
struct FoldJob { const float* ws; float* out; size_t count, stride, sub_stride; int nrec, t_cols, sub; unsigned block0, blocks; };
struct FoldJobs {
    static constexpr int MAX = 4;
    FoldJob j[MAX]; int n; unsigned total;
    FoldJobs() : n(0), total(0) {}
    // out[i] += sum_s ws[s * stride + i], i < count; t_cols as splitk_reduce_kernel.  sub > 1: every record holds `sub` rows, sub_stride
    // apart, that all fold into the SAME output (two jobs must never share an output: their workgroups run side by side).
    void add(const float* ws, float* out, int nrec, size_t stride, size_t count, int t_cols = 0, int sub = 1, size_t sub_stride = 0) {
        FoldJob& f = j[n++];
        f.ws = ws; f.out = out; f.count = count; f.stride = stride; f.nrec = nrec * sub; f.t_cols = t_cols; f.sub = sub; f.sub_stride = sub_stride;
        f.block0 = total; f.blocks = (unsigned)reduce_grid(count);
        total += f.blocks;
    }
};
static __global__ void __launch_bounds__(256)
fold_jobs_kernel(FoldJobs jobs) {
    int k = 0;
#pragma unroll
    for (int i = 1; i < FoldJobs::MAX; i++) k += (i < jobs.n && blockIdx.x >= jobs.j[i].block0) ? 1 : 0;
    const FoldJob f = jobs.j[k];
    const size_t stride = f.stride, sub_stride = f.sub_stride;
    const int sub = f.sub;
    if (sub == 1)
        reduce_rows_32x8(f.ws, f.nrec, f.count, [stride](int s) { return (size_t)s * stride; }, f.out, f.t_cols, blockIdx.x - f.block0, f.blocks);
    else
        reduce_rows_32x8(f.ws, f.nrec, f.count, [stride, sub_stride, sub](int s) { return (size_t)(s / sub) * stride + (size_t)(s % sub) * sub_stride; },
                         f.out, f.t_cols, blockIdx.x - f.block0, f.blocks);
}
inline void launch_fold_jobs(const FoldJobs& jobs, hipStream_t st) {
    if (jobs.n > 0) hipLaunchKernelGGL(fold_jobs_kernel, dim3(jobs.total), dim3(256), 0, st, jobs);
}

// conv input-gradient of one parity class: row m = (frame, yy, xx) -> pixel (s*yy+py, s*xx+px); out = v + add
template <class T> struct EpDgradScatter {
    __device__ __forceinline__ void begin_block(int) {}
    static constexpr int UNIT = 8;
    typedef EpNone Cols;
    typedef EpFragAux<T> Aux;
    T* out; const T* add; int H, W, Cin, s, py, px; FastDiv dHcWc, dWc;
    __device__ __forceinline__ size_t offset(int m, int n) const {
        uint32_t f, rem, yy, xx;
        dHcWc.divmod((uint32_t)m, f, rem);
        dWc.divmod(rem, yy, xx);
        return (((size_t)f * H + yy * s + py) * W + xx * s + px) * Cin + n;
    }
    __device__ __forceinline__ Cols cols(int, bool) const { return Cols(); }
    __device__ __forceinline__ Aux fetch(int m, int n) const {
        Aux a;
        if (add) a.f = frag_load<T>(add + offset(m, n));
        return a;
    }
    __device__ __forceinline__ void apply(int m, int n, float (&v)[8], const Aux& aux, const Cols&) const {
        float a[8]; frag_to_float<T>(aux.f, a);
#pragma unroll
        for (int i = 0; i < 8; i++) v[i] += a[i];
        frag_store<T>(out + offset(m, n), frag_from_float<T>(v));
    }
};

// ConvLSTM gates.  The 1x1-conv weight rows are pre-permuted so that GEMM column
//   n' = (c/8)*32 + gate*8 + (c%8)      (gate order f,i,o,g — reference rnn.py:57-64)
// i.e. a 32-column unit holds the four gates of 8 consecutive channels.
template <class T> struct EpLstm {
    __device__ __forceinline__ void begin_block(int) {}
    static constexpr int UNIT = 32;
    typedef EpColVec<32> Cols;
    struct Aux {
        f32x4 c0, c1;
        __device__ __forceinline__ Aux() : c0(f32x4{0.f, 0.f, 0.f, 0.f}), c1(f32x4{0.f, 0.f, 0.f, 0.f}) {}
    };
    const float* bias;      // permuted like the columns, length 4C
    const float* c_prev;    // [rows][C] fp32
    float* c_out;           // [rows][C] fp32
    T* h_out;               // [rows][C]
    T* gates;               // [rows][4C] natural layout [f|i|o|g] (activated), may be null
    int C;
    __device__ __forceinline__ Cols cols(int n, bool ok) const { return ep_load_cols<32>(bias, n, ok); }
    __device__ __forceinline__ Aux fetch(int m, int n) const {
        const float* cp = c_prev + (size_t)m * C + ((n >> 5) << 3);
        Aux a;
        a.c0 = *reinterpret_cast<const f32x4*>(cp);
        a.c1 = *reinterpret_cast<const f32x4*>(cp + 4);
        return a;
    }
    __device__ __forceinline__ void apply(int m, int n, float (&v)[32], const Aux& aux, const Cols& b) const {
        int c0 = (n >> 5) << 3;
        float f[8], ig[8], o[8], g[8], cn[8], hn[8];
        const float cp[8] = {aux.c0[0], aux.c0[1], aux.c0[2], aux.c0[3], aux.c1[0], aux.c1[1], aux.c1[2], aux.c1[3]};
#pragma unroll
        for (int i = 0; i < 8; i++) {
            f[i] = sigmoid_zb(v[i], b.v[i] * -1.4426950408889634f);          // (same forms as the scan kernels: csrc/lstm_scan.hpp)
            ig[i] = sigmoid_zb(v[8 + i], b.v[8 + i] * -1.4426950408889634f);
            o[i] = sigmoid_zb(v[16 + i], b.v[16 + i] * -1.4426950408889634f);
            g[i] = tanh_zb(v[24 + i], b.v[24 + i] * 2.8853900817779268f);
            cn[i] = f[i] * cp[i] + ig[i] * g[i];
            hn[i] = o[i] * tanh_f(cn[i]);
        }
        float* co = c_out + (size_t)m * C + c0;
        *reinterpret_cast<f32x4*>(co) = f32x4{cn[0], cn[1], cn[2], cn[3]};
        *reinterpret_cast<f32x4*>(co + 4) = f32x4{cn[4], cn[5], cn[6], cn[7]};
        frag_store<T>(h_out + (size_t)m * C + c0, frag_from_float<T>(hn));
        if (gates) {
            T* gp = gates + (size_t)m * 4 * C + c0;
            frag_store<T>(gp, frag_from_float<T>(f));
            frag_store<T>(gp + C, frag_from_float<T>(ig));
            frag_store<T>(gp + 2 * C, frag_from_float<T>(o));
            frag_store<T>(gp + 3 * C, frag_from_float<T>(g));
        }
    }
};

// ------------------------------------------------------------------------------------------------
// the kernel
// ------------------------------------------------------------------------------------------------
// LDS: operand stages (one when the whole K fits a single K tile, else two) overlaid with the fp32 epilogue staging of
// 64 rows at a time.  128x128 tile: 64 KiB double-buffered, 33.8 KiB when ONE_K (=> 3-4 workgroups per CU).
template <int BN, bool ONE_K> struct GemmSmem {
    static constexpr int MAIN = (ONE_K ? 1 : 2) * (128 + BN) * 128;
    static constexpr int EPI = 64 * (BN + 4) * 4;
    static constexpr int BYTES = MAIN > EPI ? MAIN : EPI;
};

template <class T, int BN, bool TN, bool ONE_K, class ASrc, class AXf, class BSrc, class BXf, class Ep>
__global__ void __launch_bounds__(256, (TN ? (sizeof(T) == 2 && BXf::identity ? TN_WAVES : 1) : (BN == 64 && Ep::UNIT == 8 && sizeof(T) == 2 && AXf::identity ? 3 : (BN == 128 && sizeof(T) == 4 ? 1 : 2))))
gemm_kernel(ASrc as, AXf axf, BSrc bs, BXf bxf, Ep ep, int M, int N, int K, int m_tiles, int n_tiles, int ksplit_len,
            float* a_colsum, int panel_major) {
    constexpr int BM = 128;
    constexpr int BK = TileGeom<T>::BK;
    constexpr int WN = BN / 64;
    constexpr int STAGE_BYTES = (BM + BN) * 128;   // A tile then B tile, two stages
    __shared__ __attribute__((aligned(16))) char smem[GemmSmem<BN, ONE_K>::BYTES];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    // TN (split-K) launches: workgroups are dispatched round-robin over the 8 XCDs (private L2s).  The tiles of one
    // K slice share an operand, so deal them to ids that differ by a multiple of 8 (same XCD, adjacent in time): the
    // second reader then hits L2 instead of HBM.  Pure speed: any placement is correct.
    int bx = blockIdx.x, by = blockIdx.y;
    if (TN && (gridDim.y % 8) == 0 && gridDim.x > 1) {
        const int L = blockIdx.y * gridDim.x + blockIdx.x, span = 8 * gridDim.x;
        by = (L / span) * 8 + (L % 8);
        bx = (L % span) / 8;
    }
    const int kbeg = by * ksplit_len;
    const int kend = (kbeg + ksplit_len < K) ? kbeg + ksplit_len : K;
    const int nk = (kend - kbeg + BK - 1) / BK;
    const int G = gridDim.x;

    // Persistent tile walk (NT GEMMs launch ~2-3 workgroups per CU and stride over the tiles; TN launches one
    // workgroup per (tile, K-slice)).  panel_major: a workgroup sweeps all N tiles of its 128-row panel so the
    // A panel comes from HBM once; otherwise tiles are dealt round-robin with N fastest.
    auto tile_of = [&](int seq, int& mt, int& nt) -> bool {
        if (panel_major == 2) {
            // gather sources (im2col / conv-dgrad): neighbouring output tiles re-read each other's input rows (a 7x7
            // stride-4 window shares 3 of its 7 rows with the next output row).  Workgroups are dispatched round-robin
            // over the 8 XCDs, so give every XCD one CONTIGUOUS eighth of the tile sequence and let its workgroups walk
            // it side by side: the re-reads then hit that XCD's L2 instead of going back to HBM.
            const int per = G >> 3, total = m_tiles * n_tiles, chunk = (total + 7) >> 3;
            const int w = (bx >> 3) + seq * per;
            const int t = (bx & 7) * chunk + w;
            mt = t / n_tiles; nt = t - mt * n_tiles;
            return w < chunk && t < total;
        }
        if (panel_major == 3) {
            // several N tiles per 128-row panel and too few panels for panel-major sweeps: dealing tiles N-fastest over consecutive
            // workgroup ids puts the n_tiles readers of one A panel on up to 8 XCDs, i.e. 8 private L2s each fetch it (measured on
            // M = 120960, N = 2048, K = 512: 1.04 GB fetched for a 124 MB A).  Instead panel p belongs to XCD p % 8 (workgroup id
            // mod 8) and that XCD's workgroups walk its (panel, N tile) list side by side: one fetch, then L2 hits.
            const int per = G >> 3, xcd = bx & 7;
            const int u = (bx >> 3) + seq * per;
            const int np_x = (m_tiles - xcd + 7) >> 3;
            mt = (u / n_tiles) * 8 + xcd; nt = u - (u / n_tiles) * n_tiles;
            return u < np_x * n_tiles;
        }
        if (panel_major) { mt = bx + (seq / n_tiles) * G; nt = seq % n_tiles; return mt < m_tiles; }
        const int t = bx + seq * G;
        mt = t / n_tiles; nt = t - mt * n_tiles;
        return t < m_tiles * n_tiles;
    };

    typedef typename std::conditional<TN, TNLoader<T, BM, ASrc, AXf, false>, NTLoader<T, BM, ASrc, AXf>>::type LA;
    typedef typename std::conditional<TN, TNLoader<T, BN, BSrc, BXf, true>, NTLoader<T, BN, BSrc, BXf>>::type LB;
    LA la; LB lb;
    ep.begin_block(by);

    // epilogue geometry: a thread keeps one UNIT-wide column unit and walks rows (see the epilogue protocol above)
    constexpr int LDS_LD = BN + 4;
    constexpr int UNIT = Ep::UNIT;
    constexpr int UPR = BN / UNIT;                       // units per tile row (a power of two <= 16)
    constexpr int RSTEP = 256 / UPR;                     // rows covered by the 256 threads at once
    constexpr int UPT = (64 + RSTEP - 1) / RSTEP;        // units per thread per 64-row pass
    const int ep_cu = tid % UPR, ep_row0 = tid / UPR;

    f32x16 acc[2][WN];
    auto zero_acc = [&]() {
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
            for (int j = 0; j < WN; j++) acc_zero(acc[i][j]);
    };
    // Bias gradient of a TN launch = column sums of its A operand over the K slice.  They ride on the matrix cores:
    // A_tile . ones accumulates sum_k A[row][k] in every column of a 32x32 block, so the loader does no per-element
    // work for them.  The two waves that share an A row panel (wn = 0/1) take one 32-row block each.
    constexpr bool CAN_COLSUM = TN && !BSrc::UNIT_LINEAR && !ASrc::UNIT_LINEAR;      // (conv weights have no bias: no column sums, 16 registers back)
    f32x16 colacc;
    acc_zero(colacc);
    frag_t<T> ones;
#pragma unroll
    for (int e = 0; e < 8; e++) ones[e] = (T)1.0f;
    auto mma_stage = [&](int cur, bool with_colsum) {    // acc += A_tile . B_tile of LDS operand stage `cur`
        const char* At = smem + cur * STAGE_BYTES;
        const char* Bt = At + BM * 128;
#pragma unroll
        for (int ks = 0; ks < BK / 16; ks++) {
            const int fc = ks * 2 + (lane >> 5);
            frag_t<T> a[2], b[WN];
#pragma unroll
            for (int i = 0; i < 2; i++) a[i] = tile_load_frag<T>(At, wm * 64 + i * 32 + (lane & 31), fc);
#pragma unroll
            for (int j = 0; j < WN; j++) b[j] = tile_load_frag<T>(Bt, wn * (BN / 2) + j * 32 + (lane & 31), fc);
#pragma unroll
            for (int i = 0; i < 2; i++)
#pragma unroll
                for (int j = 0; j < WN; j++) mma32(acc[i][j], a[i], b[j]);
            if (CAN_COLSUM && with_colsum) {             // workgroup-uniform
                if (wn == 0) mma32(colacc, a[0], ones);
                else mma32(colacc, a[1], ones);
            }
        }
    };
    // epilogue: accumulators -> LDS (fp32, row pitch BN+4) -> UNIT-wide row segments, 64 tile rows per pass
    // (pass i = MFMA row block i of every wave: stage row wm*32+r <-> tile row wm*64+i*32+r)
    auto side_fetch = [&](int m0, int ncol, bool col_ok, typename Ep::Cols& cols, typename Ep::Aux (&aux)[2][UPT]) {
        cols = ep.cols(ncol, col_ok);
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
            for (int q = 0; q < UPT; q++) {
                const int srow = ep_row0 + q * RSTEP;
                const int m = m0 + (srow >> 5) * 64 + i * 32 + (srow & 31);
                if (col_ok && srow < 64 && m < M) aux[i][q] = ep.fetch(m, ncol);
            }
    };
    auto epilogue = [&](int m0, int ncol, bool col_ok, const typename Ep::Cols& cols, const typename Ep::Aux (&aux)[2][UPT]) {
        float* stage = reinterpret_cast<float*>(smem);
#pragma unroll
        for (int i = 0; i < 2; i++) {
            if (i) lds_barrier();
#pragma unroll
            for (int j = 0; j < WN; j++)
#pragma unroll
                for (int r = 0; r < 16; r++)
                    stage[(wm * 32 + acc_row(r, lane)) * LDS_LD + wn * (BN / 2) + j * 32 + (lane & 31)] = acc[i][j][r];
            lds_barrier();
#pragma unroll
            for (int q = 0; q < UPT; q++) {
                const int srow = ep_row0 + q * RSTEP;
                const int m = m0 + (srow >> 5) * 64 + i * 32 + (srow & 31);
                if (col_ok && srow < 64 && m < M) {
                    float v[UNIT];
#pragma unroll
                    for (int w = 0; w < UNIT / 4; w++) {
                        f32x4 t = *reinterpret_cast<const f32x4*>(stage + srow * LDS_LD + ep_cu * UNIT + w * 4);
                        v[w * 4 + 0] = t[0]; v[w * 4 + 1] = t[1]; v[w * 4 + 2] = t[2]; v[w * 4 + 3] = t[3];
                    }
                    ep.apply(m, ncol, v, aux[i][q], cols);
                }
            }
        }
    };

    if constexpr (TN) {
        // ---- split-K weight gradient: ONE output tile per workgroup, a long K (token) loop.  One workgroup per CU, so
        // the bytes in flight ARE the prefetch depth: two K tiles (2 x 32 KiB) are kept in flight in two register sets
        // while a third is consumed from LDS. ----
        int mt, nt;
        if (!tile_of(0, mt, nt)) return;
        const int m0 = mt * BM, n0 = nt * BN;
        const int ncol = n0 + ep_cu * UNIT;
        const bool col_ok = ep_row0 < 64 && ncol < N;
        typename LA::Regs R0, R1;
        const bool want_colsum = CAN_COLSUM && a_colsum != nullptr && n0 == 0;
        zero_acc();
        la.init(as, m0, tid);
        lb.init(bs, n0, tid);
        if (nk > 0) {
            la.load(as, axf, kbeg, kend, tid, R0);
            lb.load(bs, bxf, kbeg, kend, tid, R0);
        }
        if (nk > 1) {
            la.load(as, axf, kbeg + BK, kend, tid, R1);
            lb.load(bs, bxf, kbeg + BK, kend, tid, R1);
        }
        if (nk > 0) {
            la.store(smem, axf, tid, R0);
            lb.store(smem + BM * 128, bxf, tid, R0);
        }
        lds_barrier();
        // step kt: LDS stage `cur` holds K tile kt, `Rnext` holds kt+1 (in flight), `Rfree` is reloaded with kt+2
        auto step = [&](int kt, typename LA::Regs& Rfree, typename LA::Regs& Rnext, int cur) {
            if (kt + 2 < nk) {
                la.load(as, axf, kbeg + (kt + 2) * BK, kend, tid, Rfree);
                lb.load(bs, bxf, kbeg + (kt + 2) * BK, kend, tid, Rfree);
            }
            mma_stage(cur, want_colsum);
            if (kt + 1 < nk) {
                la.store(smem + (cur ^ 1) * STAGE_BYTES, axf, tid, Rnext);
                lb.store(smem + (cur ^ 1) * STAGE_BYTES + BM * 128, bxf, tid, Rnext);
            }
            lds_barrier();
        };
        for (int kt = 0; kt < nk; kt += 2) {
            step(kt, R0, R1, 0);
            if (kt + 1 < nk) step(kt + 1, R1, R0, 1);
        }
        // bias gradient: this K slice's partial column sums of the A operand -> a_colsum[slice][M]; every column of
        // colacc holds the same sums, lanes 0 and 32 own the 2 x 16 rows of the block
        if (want_colsum && (lane & 31) == 0) {
            float* cs_out = a_colsum + (size_t)by * M;
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int row = m0 + wm * 64 + wn * 32 + acc_row(r, lane);
                if (row < M) cs_out[row] = colacc[r];
            }
        }
        typename Ep::Cols cols;
        typename Ep::Aux aux[2][UPT];
        side_fetch(m0, ncol, col_ok, cols, aux);
        epilogue(m0, ncol, col_ok, cols, aux);
    } else {
        int seq = 0, mt, nt;
        bool have = tile_of(0, mt, nt);
        if (have && nk > 0) {
            la.init(as, mt * BM, tid);
            lb.init(bs, nt * BN, tid);
            la.load(as, axf, kbeg, kend, tid);
            lb.load(bs, bxf, kbeg, kend, tid);
            la.store(smem, axf, tid);
            lb.store(smem + BM * 128, bxf, tid);
        }
        // The prefetched operands of the NEXT tile are moved to LDS at the BOTTOM of the loop body, not at its top: there
        // the compiler sees "8 loads, then this tile's output stores" on every path and waits with vmcnt(#stores).  With
        // the move at the loop top the header merges the prologue path (loads only) with the back edge, the wait becomes
        // vmcnt(0), and every tile stalls until its predecessor's output stores are acknowledged by memory.
        while (have) {
            const int m0 = mt * BM, n0 = nt * BN;
            const int ncol = n0 + ep_cu * UNIT;
            const bool col_ok = ep_row0 < 64 && ncol < N;
            zero_acc();

            // Global-memory traffic of a tile, in issue order (vmcnt counts loads and stores in ONE in-order queue, so a
            // wait for some load also waits for everything issued before it):
            //   1. side inputs of this tile's epilogue (bias/gamma columns, residual, GELU', c_prev ...)
            //   2. the operand prefetch of the NEXT tile
            //   3. this tile's output stores
            // so the epilogue arithmetic waits for (1) only, and the move of (2) into LDS at the loop bottom waits with
            // vmcnt(#stores) and never for a store acknowledgement.  (1) and (2) are issued in front of the LAST K tile's
            // MFMA phase (the K loop's own operand loads share the loader registers until then): a full tile time ahead
            // of their use when the contraction is a single K tile, one MFMA phase plus the epilogue otherwise.
            typename Ep::Cols cols;
            typename Ep::Aux aux[2][UPT];
            int mt2 = 0, nt2 = 0;
            bool have2 = false;
            auto prefetch_next = [&]() {
                have2 = tile_of(seq + 1, mt2, nt2);
                if (have2 && nk > 0) {
                    la.init(as, mt2 * BM, tid);
                    lb.init(bs, nt2 * BN, tid);
                    la.load(as, axf, kbeg, kend, tid);
                    lb.load(bs, bxf, kbeg, kend, tid);
                }
            };

            lds_barrier();
            for (int kt = 0; kt + 1 < nk; kt++) {            // all K tiles but the last: stream the next one in
                const int cur = kt & 1;
                la.load(as, axf, kbeg + (kt + 1) * BK, kend, tid);
                lb.load(bs, bxf, kbeg + (kt + 1) * BK, kend, tid);
                mma_stage(cur, false);
                la.store(smem + (cur ^ 1) * STAGE_BYTES, axf, tid);
                lb.store(smem + (cur ^ 1) * STAGE_BYTES + BM * 128, bxf, tid);
                lds_barrier();
            }
            // last K tile (peeled, so that this is straight-line code): the loader registers are free from here on
            side_fetch(m0, ncol, col_ok, cols, aux);
            sched_fence();
            prefetch_next();
            sched_fence();      // the prefetch must be ISSUED here, not sunk below the MFMA phase / epilogue
            if (nk > 0) {
                mma_stage((nk - 1) & 1, false);
                lds_barrier();
            }

            epilogue(m0, ncol, col_ok, cols, aux);
            lds_barrier();            // staging buffer is reused by the next tile's operand stores
            have = have2; mt = mt2; nt = nt2; seq++;
            if (have && nk > 0) {
                la.store(smem, axf, tid);
                lb.store(smem + BM * 128, bxf, tid);
            }
        }
    }
}


inline int gemm_slices(int K, int ksplit, int BK) {
    if (ksplit < 1) ksplit = 1;
    int klen = ((K + ksplit - 1) / ksplit + BK - 1) / BK * BK;
    if (klen < BK) klen = BK;
    int nsplit = (K + klen - 1) / klen;
    return nsplit < 1 ? 1 : nsplit;
}

template <class T, int BN, bool TN, class ASrc, class AXf, class BSrc, class BXf, class Ep>
inline void launch_gemm(const ASrc& as, const AXf& axf, const BSrc& bs, const BXf& bxf, const Ep& ep,
                        int M, int N, int K, int ksplit, hipStream_t stream, float* a_colsum = nullptr) {
    const int BK = TileGeom<T>::BK;
    int n_tiles = (N + BN - 1) / BN;
    int m_tiles = (M + 127) / 128;
    if (ksplit < 1) ksplit = 1;
    int klen = ((K + ksplit - 1) / ksplit + BK - 1) / BK * BK;
    if (klen < BK) klen = BK;
    int nsplit = gemm_slices(K, ksplit, BK);
    int total = m_tiles * n_tiles;
    int gx = total, panel_major = 0;
    const bool one_k = !TN && K <= BK;
    if (!TN) {                                   // persistent: exactly one resident wave of workgroups striding over the tiles
        const int resident_override = g_tuning.gemm_resident;
        // workgroups per CU as the hardware will actually schedule them (registers and LDS of THIS instantiation): more
        // would queue behind the resident ones and run as a half-empty second wave
        static int per_cu_one = 0, per_cu_multi = 0;
        int& per_cu = one_k ? per_cu_one : per_cu_multi;
        if (per_cu == 0) {
#ifdef RVT_EMU
            per_cu = 2;
#else
            int nb = 0;
            hipError_t e = one_k
                ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, gemm_kernel<T, BN, TN, true, ASrc, AXf, BSrc, BXf, Ep>, 256, 0)
                : hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, gemm_kernel<T, BN, TN, false, ASrc, AXf, BSrc, BXf, Ep>, 256, 0);
            per_cu = (e == hipSuccess && nb > 0) ? nb : 2;
#endif
        }
        const int resident = resident_override > 0 ? resident_override : 256 * per_cu;
        if (total > resident) gx = resident;
        panel_major = (n_tiles > 1 && m_tiles >= 4 * gx) ? 1 : 0;
        if (ASrc::SPATIAL_REUSE && total > gx && (gx & 7) == 0) panel_major = 2;
        const int xcd_panels = g_tuning.gemm_xcd_panels;    // (A/B knob)
        if (panel_major == 0 && n_tiles > 1 && xcd_panels && total >= 8 * gx) {
            // XCD-grouped panels (tile_of, mode 3).  Only for long tile walks: the per-XCD lists differ by up to one panel, which
            // on a one- or two-round launch (the per-step ConvLSTM GEMMs: measured +16 % on rvt_lstm_dgrad) is a whole extra round.
            gx = (gx + 7) & ~7;
            panel_major = 3;
        }
    }
    if (one_k)                // whole contraction in one K tile: single operand stage, more workgroups per CU
        hipLaunchKernelGGL((gemm_kernel<T, BN, TN, true, ASrc, AXf, BSrc, BXf, Ep>), dim3(gx, nsplit), dim3(256), 0, stream,
                           as, axf, bs, bxf, ep, M, N, K, m_tiles, n_tiles, klen, a_colsum, panel_major);
    else
        hipLaunchKernelGGL((gemm_kernel<T, BN, TN, false, ASrc, AXf, BSrc, BXf, Ep>), dim3(gx, nsplit), dim3(256), 0, stream,
                           as, axf, bs, bxf, ep, M, N, K, m_tiles, n_tiles, klen, a_colsum, panel_major);
}

}  // namespace rvt
