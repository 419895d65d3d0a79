// Shared pieces of the partitioned multi-head self-attention kernels (reference maxvit.py:343-354 on the partitions of
// maxvit.py:273-304): partition geometry / token addressing, operand-chunk loads, the in-lane column softmax.  The kernels
// themselves are in attn_core2.hpp (attention core on a saved qkv tensor) and attn_block.hpp (fused attention half).
//
// The qkv activations stay in IMAGE token order [F*H*W][3C] (per-head channel layout [q|k|v], dh each — reference
// maxvit.py:347); window / grid partitioning is pure index arithmetic on the token rows that a partition gathers, so the
// reference's four permute().contiguous() copies per block never exist.
//
// MFMA formulation (L <= 96 tokens per partition, dh <= 32):
//   S^T = K Q^T        A = K rows (keys j), B = Q rows (queries i)  -> lane owns ONE query column i, its 16*NB accumulator
//                      registers run over keys  => softmax max / sum are in-lane reductions plus one exchange with lane^32.
//   O^T = V^T P^T      B = P^T straight from the accumulator registers: the MFMA pairs A's and B's k-slots one-to-one, so the
//                      keys are enumerated in the order the lane already holds them ( j = 32bj+16q+8(e>>2)+4*half+(e&3) ).
#pragma once
#include "common.hpp"

namespace rvt {

// LDS hand-off between the lanes of ONE wave (private slice): DS operations of a wave execute in issue order, so draining
// lgkmcnt (plus a compiler barrier) is all the synchronisation there is to do — no s_barrier across the workgroup.
__device__ __forceinline__ void wave_lds_sync() {
#ifdef RVT_EMU
    emu::wave_barrier();          // emulator: lanes are fibers; the LDS hand-off is between the lanes of this wave only
#else
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
}

struct AttnGeom {
    int F, H, W, C, dh, heads, ph, pw, L, window;   // L = ph*pw
    int nPw;        // partitions along x
    int P;          // partitions per frame
    float scale;
    FastDiv dGroups, dP, dnPw, dpw;   // dGroups: head groups (of HG heads) per partition
};

// image-order token row of slot l of partition p of frame f
__device__ __forceinline__ int attn_token(const AttnGeom& g, int f, int p, int l) {
    uint32_t py, px, ly, lx;
    g.dnPw.divmod((uint32_t)p, py, px);
    g.dpw.divmod((uint32_t)l, ly, lx);
    int y, x;
    if (g.window) { y = py * g.ph + ly; x = px * g.pw + lx; }                  // maxvit.py:273-279
    else { y = ly * (g.H / g.ph) + py; x = lx * (g.W / g.pw) + px; }           // maxvit.py:290-296 (dilated grid)
    return (f * g.H + y) * g.W + x;
}

template <class T> __device__ __forceinline__ frag_t<T> load_chunk(const T* row, int chunk, int dh, bool valid) {
    // branch-free: padded rows point at token 0 (always mapped); the select zeroes what is not real
    const bool ok = valid & (chunk * 8 < dh);
    const frag_t<T> v = frag_load<T>(row + (ok ? chunk * 8 : 0));
    const frag_t<T> z = frag_zero<T>();
    return ok ? v : z;
}

// store 4 consecutive channels of one token row as ONE 8-byte (bf16) / 16-byte (f32) write
template <class T> __device__ __forceinline__ void store4(T* dst, float a, float b, float c, float d) {
    typedef __attribute__((ext_vector_type(4))) T vec4;
    vec4 v;
    v[0] = (T)a; v[1] = (T)b; v[2] = (T)c; v[3] = (T)d;
    *reinterpret_cast<vec4*>(dst) = v;
}

template <class T> __device__ __forceinline__ frag_t<T> acc_slot_frag(const f32x16& a, int q) {
    frag_t<T> f;
#pragma unroll
    for (int e = 0; e < 8; e++) f[e] = (T)a[8 * q + e];
    return f;
}

template <class T> __device__ __forceinline__ frag_t<T> arr_slot_frag(const float (&a)[16], int q) {
    frag_t<T> f;
#pragma unroll
    for (int e = 0; e < 8; e++) f[e] = (T)a[8 * q + e];
    return f;
}

}  // namespace rvt
