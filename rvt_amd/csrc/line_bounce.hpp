// Row hand-over helpers of the chained ("T-form": lane = token) kernels: exact-range buffer resources for a wave's 32-row tile and
// full-line stores through a per-wave LDS bounce.
#pragma once
#include "common.hpp"
#include "ppgemm.hpp"

namespace rvt {

// wave-level rendezvous for data handed between LANES of one wave through LDS.  The hardware needs nothing (the LDS operations of a
// wave execute in order), the COMPILER does: with an empty pp_wave_sync() hipcc sank the read-back of the bounce tile below into the
// divergent `if` that guards the writes - lanes outside the branch then stored stale registers (caught by the GPU parity test only:
// the emulator's rendezvous is a real one).  Wavefront-scope fences + the convergent wave barrier pin the order; no instruction is emitted.
__device__ __forceinline__ void wave_rendezvous() {
#ifdef RVT_EMU
    emu::wave_barrier();
#else
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#endif
}

// Rows leave as FULL 128-byte lines.  In the T-form a lane owns a token, so a store of accumulator pieces writes 32 bytes of 32
// different lines; with that pattern three different kernels for fc1 + GELU (GEMM engine, streamed weights, weight-stationary) all
// sat at 2.7 - 2.9 TB/s (a contiguous fill reaches 6.7 TB/s on this part) and the first cut of lnlin_fwd_kernel at 4.6.  The pieces of a
// 64-column group (one line per token) bounce through a 2-KiB per-wave LDS tile, 16 tokens at a time, and come back as
// lane = (token lane / 8, 16-byte piece lane % 8): one store instruction = 8 whole lines.  It pays where the kernel is bound by the
// memory system (ln_linear 0.54 -> 0.46 ms, fc1 + GELU 0.82 -> 0.55, stem forward 1.56 -> 1.45); in the VALU-bound fused attention
// forward the same bounce (row indices sent to the storing lanes by a wave shuffle) COST 0.1 ms per launch and was removed again.
template <int TOK = 16> struct LineBounceT {              // TOK tokens per phase: 16 (2 KiB per wave) or 8 (1 KiB, where LDS is short)
    static_assert(TOK == 8 || TOK == 16, "8 or 16 tokens per phase");
    static constexpr int BYTES = TOK * 128;              // per wave: [TOK tokens][128 B], piece position ^ (token & 7)
    char* scr;
    int li, half, wr, wr_sw, rd_t, rd_q;
    __device__ __forceinline__ void init(char* s, int lane) {
        scr = s; li = lane & 31; half = lane >> 5;
        wr = (li % TOK) * 128; wr_sw = li & 7;
        rd_t = lane >> 3; rd_q = lane & 7;
    }
    // pc[j][m] = this lane's token, columns 32 j + 16 m + 8 half .. + 7 of the group; the group starts col_bytes into a row of row_bytes
    __device__ __forceinline__ void flush(const pp_rsrc& dst, const u32x4 (&pc)[2][2], int row_bytes, int col_bytes) const {
#pragma unroll
        for (int ph = 0; ph < 32 / TOK; ph++) {            // tokens TOK ph .. TOK ph + TOK - 1
            if (li / TOK == ph) {
#pragma unroll
                for (int j = 0; j < 2; j++)
#pragma unroll
                    for (int m = 0; m < 2; m++) *reinterpret_cast<u32x4*>(scr + wr + (((4 * j + 2 * m + half) ^ wr_sw) << 4)) = pc[j][m];
            }
            wave_rendezvous();
#pragma unroll
            for (int it = 0; it < TOK / 8; it++) {
                const int t = rd_t + 8 * it;
                const u32x4 v = *reinterpret_cast<const u32x4*>(scr + t * 128 + ((rd_q ^ (t & 7)) << 4));
                pp_store16(dst, (TOK * ph + t) * row_bytes + col_bytes + rd_q * 16, v);
            }
            wave_rendezvous();
        }
    }
};
typedef LineBounceT<16> LineBounce;

// rows of tile t that exist (0 .. 32), in a scalar register (hipcc clamps with a VALU med3, and a resource word in a vector register
// costs a waterfall loop per access)
__device__ __forceinline__ int tile_rows(int M, int t) {
    const int r = M - t * 32;
    return wave_uniform(r < 0 ? 0 : (r > 32 ? 32 : r));
}
// buffer resource over exactly the existing rows of tile t of a row-major [M][ld] matrix: rows beyond M load zeros and their stores
// are dropped, so a tile loop needs no branch and no select on a loaded value - and hipcc can COUNT the accesses behind a prefetch
// (loads and stores retire through one in-order counter: behind a conditional store it waits for vmcnt(small) = every store of the tile)
template <class T> __device__ __forceinline__ pp_rsrc tile_rsrc(const T* base, int ld, int M, int t, bool present = true) {
    return pp_make_rsrc(base + (size_t)t * 32 * ld, present ? (unsigned)(tile_rows(M, t) * ld * (int)sizeof(T)) : 0u);
}

}  // namespace rvt
