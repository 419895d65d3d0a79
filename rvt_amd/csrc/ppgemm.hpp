// 256 x 256 x 64 bf16 GEMM for the MFMA-bound contractions of the RVT hot path (stages 3-4 linears and their input
// gradients, the per-step ConvLSTM products):
//
//     D[m][n] = sum_k X[m][k] * W[n][k]            X = token rows, W = weight rows, both "row, k" with k contiguous
//
// gemm.hpp's 128 x 128 / 4-wave engine stages its operands through registers (one K tile in flight, one ds_read_b128 per
// MFMA) and reaches 19-27 % of the MFMA peak on these shapes.  This kernel is built around the LDS-DMA path instead:
//
//   * one 512-thread workgroup per CU, 8 waves as 2 (token halves of 128 rows) x 4 (64 weight rows each); a wave owns a
//     128 x 64 block of the tile = 4 x 2 MFMA 32x32 blocks (128 accumulator registers): 24 ds_read_b128 per 32 MFMAs;
//   * operands go global -> LDS with buffer_load_dwordx4 ... lds (no loader registers): a K tile is four 16 KiB
//     "half tiles" (X0, W0, W1, X1: the token / weight rows of one quadrant of every wave), two LDS stages of 64 KiB;
//     the LDS image is lane-linear per wave instruction, so the bank swizzle (16-byte chunk ^= (row >> 1) & 7) is applied to
//     the per-lane SOURCE address and to the fragment reads;
//   * the load stream runs 6 half tiles (1.5 K tiles) ahead of the MFMAs and never drains: every phase issues one half
//     tile (2 instructions per wave) and waits with a COUNTED vmcnt (4 half tiles stay in flight); the stream is
//     indexed by the workgroup's K-tile step, not by output tile, so it runs straight across tile boundaries;
//   * the two wave groups (token half 0 / 1 = the two waves of every SIMD) run one barrier apart: while one group issues
//     its 8 MFMAs of a quadrant, the other reads fragments, issues LDS-DMA and - during the first K tile of the next
//     output tile - converts and stores the finished quadrant of the previous tile ("ping-pong");
//   * T-form products (A operand = weight rows, B operand = token rows): a lane holds 16 features of ONE token per
//     accumulator block, so the epilogue works on 8 consecutive columns of a row (16 bytes) after one v_permlane32_swap per
//     pair of registers - no LDS staging of the result; per-column constants (bias, LayerScale) sit in LDS;
//   * rows beyond M are never loaded and never stored: every global access goes through a buffer descriptor that ends at
//     the tile's last valid row (out-of-range loads return zeros, out-of-range stores are dropped) - the instructions are
//     issued regardless, which keeps the vmcnt bookkeeping below exact.
//
// Hazard bookkeeping (half tile with sequence number n = 4 step + {X0:0, W0:1, W1:2, X1:3}; phase k = 4 step + p):
//   issue   n is issued in phase n - 6;
//   RAW     phase k waits (after its own issue) until everything up to n = k + 2 has landed, then passes a barrier; n is
//           first read in phase >= n - 1 > its wait phase.  vmcnt retires loads AND stores in issue order on gfx9, so the
//           count is "vector memory instructions younger than half tile k + 2": the 8 LDS-DMA pieces of four half tiles plus
//           the epilogue's stores / side loads issued since (EPI_OPS per flushed quadrant; step_body's W0..W3);
//   WAR     n overwrites n - 8, last read in phase <= n - 8 by fragment reads that have returned before that phase's MFMAs,
//           i.e. two barriers before any wave issues n.
#pragma once
#include "common.hpp"
#include "gemm.hpp"

namespace rvt {

struct PPMat {            // row-major bf16 operand whose K axis may be cut in two segments ([x_t | h_{t-1}], rnn.py:52)
    const bf16* p0; const bf16* p1; int ld; int kcut;        // column k < kcut: p0[r * ld + k], else p1[r * ld + k - kcut]
};

// epilogues (reference: maxvit.py:347,353,106,113 bias; :51-53,268-269 LayerScale + residual; autograd of the same)
enum { PP_STORE = 0,        // out = v + c0[n]
       PP_SCALE_RES = 1,    // out = side[m][n] + c1[n] * (v + c0[n])
       PP_GELU_DUAL = 2,    // out = gelu(v + c0[n]), out2 = gelu'(v + c0[n])   (out2 nullable)
       PP_MUL = 3,          // out = v * side[m][n]
       PP_ADD = 4 };        // out = v + c0[n] + side[m][n]
struct PPEpArgs {
    bf16* out; bf16* out2; const bf16* side; const float* c0; const float* c1; int ld;
};

#ifdef RVT_EMU
struct pp_rsrc { const char* base; unsigned bytes; };
__device__ __forceinline__ pp_rsrc pp_make_rsrc(const void* base, unsigned bytes) { return pp_rsrc{(const char*)base, bytes}; }
// one LDS-DMA piece: LDS[lds_off + 16 lane ..) = mem[voff + soff ..), zeros when out of range
__device__ __forceinline__ void pp_glds16(const pp_rsrc& rs, char* smem, int lds_off, int voff, int soff) {
    char* d = smem + lds_off + 16 * emu::g.cur->lane;
    const size_t o = (size_t)(unsigned)voff + (size_t)(unsigned)soff;
    if (o + 16 <= rs.bytes) memcpy(d, rs.base + o, 16);
    else memset(d, 0, 16);
}
__device__ __forceinline__ u32x4 pp_load16(const pp_rsrc& rs, int voff) {
    u32x4 v = {0u, 0u, 0u, 0u};
    if ((size_t)(unsigned)voff + 16 <= rs.bytes) memcpy(&v, rs.base + (unsigned)voff, 16);
    return v;
}
__device__ __forceinline__ void pp_store16(const pp_rsrc& rs, int voff, const u32x4& v) {
    if ((size_t)(unsigned)voff + 16 <= rs.bytes) memcpy(const_cast<char*>(rs.base) + (unsigned)voff, &v, 16);
}
template <int N> __device__ __forceinline__ void pp_wait_vm() {}
__device__ __forceinline__ void pp_barrier() { __syncthreads(); }
__device__ __forceinline__ void pp_wave_sync() { emu::wave_barrier(); }
__device__ __forceinline__ void pp_setprio(int) {}
#else
typedef __amdgpu_buffer_rsrc_t pp_rsrc;
__device__ __forceinline__ pp_rsrc pp_make_rsrc(const void* base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ void pp_glds16(const pp_rsrc& rs, char* smem, int lds_off, int voff, int soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(smem + lds_off), 16, voff, soff, 0, 0);
}
__device__ __forceinline__ u32x4 pp_load16(const pp_rsrc& rs, int voff) { return __builtin_amdgcn_raw_buffer_load_b128(rs, voff, 0, 0); }
__device__ __forceinline__ void pp_store16(const pp_rsrc& rs, int voff, const u32x4& v) { __builtin_amdgcn_raw_buffer_store_b128(v, rs, voff, 0, 0); }
template <int N> __device__ __forceinline__ void pp_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void pp_barrier() { __builtin_amdgcn_s_barrier(); }
__device__ __forceinline__ void pp_wave_sync() {}
#define pp_setprio(x) __builtin_amdgcn_s_setprio(x)
#endif

struct PPGeom {
    static constexpr int BM = 256, BN = 256, BK = 64;
    static constexpr int UNIT = 64 * 128;             // bytes of a load-stream unit: 64 rows x 128 B = one LDS-DMA instruction per wave
    static constexpr int STAGE = 8 * UNIT;            // X0 X1 X2 X3 | W0 W1 W2 W3
    static constexpr int OFF_W = 4 * UNIT;
    static constexpr int MAX_CST = 2048;              // per-column constants in LDS: MAX_CST floats (c0 [, c1])
    static constexpr int OFF_CST = 2 * STAGE;
    static constexpr int SCR_LD = 144;                // epilogue scratch: 32 rows x 128 B of bf16, rows padded to 144 B
    static constexpr int SCR_BYTES = 32 * SCR_LD;     // per wave pair (w, w + 4): the two never flush in the same interval
    static constexpr int OFF_SCR = OFF_CST + MAX_CST * 4;
    static constexpr int SMEM = OFF_SCR + 4 * SCR_BYTES;
};

// Work list of a workgroup: 256-row panel p belongs to XCD p % 8 (workgroup id % 8: the dispatcher deals workgroups
// round-robin over the 8 XCDs) and that XCD's workgroups walk its (panel, N tile) list side by side, N fastest - the
// readers of one X panel share an L2.  Pure speed: any placement computes the same tiles.
struct PPWork {
    int xcd, slot, per, n_tiles, count;
    __device__ __forceinline__ void init(int bid, int grid, int m_tiles, int n_tiles_) {
        xcd = bid & 7; slot = bid >> 3; per = grid >> 3; n_tiles = n_tiles_;
        const int items = ((m_tiles - xcd + 7) >> 3) * n_tiles;
        count = slot < items ? (items - slot + per - 1) / per : 0;
    }
    __device__ __forceinline__ void tile(int u, int& mt, int& nt) const {
        const int idx = slot + u * per, q = idx / n_tiles;
        mt = q * 8 + xcd; nt = idx - q * n_tiles;
    }
};

// GATHER = 1: input gradient of the 3 x 3 / stride 2 / pad 1 down-sampling conv (reference maxvit.py:160-168, autograd) as ONE
// product.  Row m = (frame, a, b) = the 2 x 2 block of input pixels (2a + py, 2b + px); its four pixels only see the four dY
// pixels (a + da, b + db), da, db in {0, 1} (ky = 1 | 2, 0 for py = 0 | 1 and da = 0, 1; the same in x):
//     [dIn(2a+py, 2b+px)]_{py,px}  =  [dY(a+da, b+db)]_{da,db}  .  Wd4^T,    Wd4[(py,px,ci)][(da,db,co)] = w[co][ci][ky][kx] or 0
// The X operand is the dY tensor itself: the K tiles of tap (da, db) read row m + da Wo + db (one scalar offset per step; lanes
// whose tap leaves the image or whose row is beyond M get an out-of-range address = zeros), the output row is scattered to its
// four pixels (wave column -> (py, px, channel block)).  7 of the 16 (class, tap) blocks of Wd4 are zero: an N tile only walks the
// taps one of its classes uses (taps[nt] bit mask): all 16 / 16 for Cin = 64, 12 / 16 for Cin = 128, 9 / 16 for Cin = 256.
struct PPConv {
    int Ho, Wo, Cout, H, W, Cin;          // dY [F][Ho][Wo][Cout], dIn [F][H][W][Cin]
    FastDiv dHoWo, dWo;
    int taps[8];                          // per N tile (N = 4 Cin <= 2048: up to 8 tiles): bit (2 da + db)
};

// GATHER = 2 (round 6): the 3 x 3 / stride 2 / pad 1 down-sampling conv itself (reference maxvit.py:160-168) with im2col in the load
// stream: row m = output pixel (frame, oy, ox), K tile (tap (ky, kx), 64-channel chunk) reads the input pixel
// (2 oy - 1 + ky, 2 ox - 1 + kx) - a per-lane source address formed when the unit is issued (two exact divisions by constants;
// the load side runs ahead of the output tile, so nothing per tile is kept in registers), out-of-image taps (oy = 0 with ky = 0,
// ox = 0 with kx = 0; H, W even) and rows beyond M get an out-of-range address = zeros.  W = the tap-major packed weight
// [Cout][9 Cin] (PACK_CONV_FWD), plain output rows.  PPConv: Ho, Wo = output size, H, W = input size, Cout = channels per tap (Cin).
// ABL: ablation bits for profiles/probes/ppgemm_probe.hip (0 in the library): 1 no LDS-DMA / vmcnt waits, 2 no fragment reads,
// 4 no MFMAs, 8 no barriers, 16 no epilogue, 32 epilogue stores dropped (empty output buffer)
template <int EP, int ABL = 0, int GATHER = 0>
__global__ void __launch_bounds__(512, 2)
ppgemm_kernel(PPMat X, PPMat W, PPEpArgs ep, int M, int N, int K, int m_tiles, int n_tiles, PPConv cv) {
    typedef PPGeom G;
    __shared__ __attribute__((aligned(16))) char smem[G::SMEM];
    float* const cst = reinterpret_cast<float*>(smem + G::OFF_CST);
    const int tid = threadIdx.x, lane_ = tid & 63, lane = lane_, l31 = lane & 31, hi = lane >> 5;
    const int wave = wave_uniform(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int cpt = GATHER ? cv.Cout / G::BK : 0;     // K tiles per tap
    auto nk_of = [&](int nt) __attribute__((always_inline)) { return GATHER == 1 ? __builtin_popcount(cv.taps[nt & 7]) * cpt : K / G::BK; };
    constexpr bool SIDE = EP == PP_ADD || EP == PP_SCALE_RES || EP == PP_MUL;

    PPWork work;
    work.init(blockIdx.x, gridDim.x, m_tiles, n_tiles);
    if (work.count == 0) return;
    int nsteps = 0;
    if (GATHER == 1) {
        for (int u = 0; u < work.count; u++) { int mt_, nt_; work.tile(u, mt_, nt_); nsteps += nk_of(nt_); }
    } else {
        nsteps = work.count * nk_of(0);
    }

    // per-column constants -> LDS (before the load stream starts: these are ordinary loads)
    const int c1off = EP == PP_SCALE_RES ? N : 0;
    for (int i = tid; i < N; i += 512) {
        cst[i] = ep.c0 ? ep.c0[i] : 0.f;
        if (EP == PP_SCALE_RES) cst[c1off + i] = ep.c1[i];
    }

    // ---- load stream.  Unit = 64 tile rows x 128 B, one LDS-DMA instruction per wave (8 rows x 128 B per wave).  X unit i =
    // MFMA row block i of both wave groups (tile rows wr * 128 + i * 32 + r), W unit g = the 64 weight rows of wave column g.
    // The LDS image is lane-linear, so the bank swizzle (chunk ^= (row >> 1) & 7) goes on the per-lane SOURCE chunk. ----
    const int lr = wave * 8 + (lane >> 3);                              // row of the unit this lane fills
    const int chunk = (lane & 7) ^ ((lr >> 1) & 7);
    const int vx0 = ((lr >> 5) * 128 + (lr & 31)) * X.ld * 2 + chunk * 16;      // + i * 32 rows
    const int vw0 = lr * W.ld * 2 + chunk * 16;                                  // + g * 64 rows
    const int ldx64 = X.ld * 64, ldw128 = W.ld * 128;                   // 32 rows of X / 64 rows of W, in bytes
    const int lds0 = wave * 1024;
    // ---- fragment reads: k-step ks sits at chunk (2 ks + hi) ^ swz = chunk(0) ^ 2 ks ----
    const int swz = (l31 >> 1) & 7;
    const int xaddr0 = (wr * 32 + l31) * 128 + ((hi ^ swz) << 4);                // + i * UNIT
    const int waddr0 = G::OFF_W + wc * G::UNIT + l31 * 128 + ((hi ^ swz) << 4);  // + j * 32 rows

    // descriptors of the steps s+1 and s+2
    struct Desc { pp_rsrc rx, rw; int sx, sw, da, db, mask; };
    int lu = 0, lkt = 0, lstep = 0;                  // load-side (item, K tile, step)
    int gmask = 0;                                   // GATHER, current load-side tile: bit i: row of unit i is beyond M; 4 + i: a == Ho - 1; 8 + i: b == Wo - 1
    auto next_desc = [&]() -> Desc {
        Desc d;
        d.da = 0; d.db = 0; d.mask = 0;
        if (lstep < nsteps) {
            int mt, nt;
            work.tile(lu, mt, nt);
            if (GATHER == 2) {
                const int tap = lkt / cpt, coff = (lkt - tap * cpt) * G::BK;
                const int ky = tap / 3;
                d.da = ky; d.db = tap - 3 * ky; d.mask = mt * G::BM;              // (mask: first row of the tile)
                d.rx = pp_make_rsrc(X.p0, (unsigned)((size_t)(M / (cv.Ho * cv.Wo)) * cv.H * cv.W * cv.Cout * 2));
                d.sx = coff * 2;
                d.rw = pp_make_rsrc(W.p0 + (size_t)nt * G::BN * W.ld, (unsigned)(G::BN * W.ld * 2));
                d.sw = lkt * G::BK * 2;
            } else if (GATHER) {
                if (lkt == 0) {                      // new tile on the load side: border / tail masks of this lane's four rows
                    gmask = 0;
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        const int m = mt * G::BM + (lr >> 5) * 128 + i * 32 + (lr & 31);
                        uint32_t f, rem, a, b;
                        cv.dHoWo.divmod((uint32_t)(m < M ? m : 0), f, rem);
                        cv.dWo.divmod(rem, a, b);
                        gmask |= (m >= M ? 1 : 0) << i;
                        gmask |= ((int)a == cv.Ho - 1 ? 1 : 0) << (4 + i);
                        gmask |= ((int)b == cv.Wo - 1 ? 1 : 0) << (8 + i);
                    }
                }
                const int tidx = lkt / cpt, coff = (lkt - tidx * cpt) * G::BK;
                int tap = 0, cnt = 0;
#pragma unroll
                for (int t = 0; t < 4; t++) {        // the tidx-th tap of this N tile
                    const int on = (cv.taps[nt & 7] >> t) & 1;
                    if (on && cnt == tidx) tap = t;
                    cnt += on;
                }
                d.da = tap >> 1; d.db = tap & 1; d.mask = gmask;
                d.rx = pp_make_rsrc(X.p0, (unsigned)((size_t)M * X.ld * 2));
                d.sx = (mt * G::BM * X.ld + (d.da * cv.Wo + d.db) * cv.Cout + coff) * 2;
                d.rw = pp_make_rsrc(W.p0 + (size_t)nt * G::BN * W.ld, (unsigned)(G::BN * W.ld * 2));
                d.sw = (tap * cv.Cout + coff) * 2;
            } else {
                const int k0 = lkt * G::BK;
                const bool seg = k0 >= X.kcut;
                const bf16* xb = (seg ? X.p1 : X.p0) + (size_t)mt * G::BM * X.ld;
                const int rows = M - mt * G::BM;
                d.rx = pp_make_rsrc(xb, (unsigned)((rows > G::BM ? G::BM : rows) * X.ld * 2));
                d.sx = (seg ? k0 - X.kcut : k0) * 2;
                d.rw = pp_make_rsrc(W.p0 + (size_t)nt * G::BN * W.ld, (unsigned)(G::BN * W.ld * 2));
                d.sw = k0 * 2;
            }
            if (++lkt == nk_of(nt)) { lkt = 0; lu++; }
        } else {                                     // past the end: empty buffers (zeros to LDS, no memory traffic) keep the counts uniform
            d.rx = pp_make_rsrc(X.p0, 0u); d.rw = pp_make_rsrc(W.p0, 0u); d.sx = 0; d.sw = 0;
        }
        lstep++;
        return d;
    };
    auto issue_x = [&](const Desc& d, int stage, int i) __attribute__((always_inline)) {
        int v = vx0 + i * ldx64;
        if (GATHER == 2) {
            const int m = d.mask + (lr >> 5) * 128 + i * 32 + (lr & 31);
            uint32_t f, rem, oy, ox;
            cv.dHoWo.divmod((uint32_t)(m < M ? m : 0), f, rem);
            cv.dWo.divmod(rem, oy, ox);
            const int pix = ((int)f * cv.H + 2 * (int)oy - 1 + d.da) * cv.W + 2 * (int)ox - 1 + d.db;
            const bool kill = m >= M || ((int)oy == 0 && d.da == 0) || ((int)ox == 0 && d.db == 0);
            v = kill ? (int)0x80000000u : pix * cv.Cout * 2 + chunk * 16;
        } else if (GATHER) {
            const bool kill = ((d.mask >> i) & 1) | (d.da & (d.mask >> (4 + i)) & 1) | (d.db & (d.mask >> (8 + i)) & 1);
            v = kill ? (int)0x80000000u : v;         // beyond any permitted extent (the descriptor spans < 2^31 bytes), whatever the scalar offset: the DMA writes zeros
        }
        if (!(ABL & 1)) pp_glds16(d.rx, smem, stage * G::STAGE + i * G::UNIT + lds0, v, d.sx);
    };
    auto issue_w = [&](const Desc& d, int stage, int g) __attribute__((always_inline)) {
        if (!(ABL & 1)) pp_glds16(d.rw, smem, stage * G::STAGE + G::OFF_W + g * G::UNIT + lds0, vw0 + g * ldw128, d.sw);
    };

    f32x16 acc[4][2];
    bf16x8 xf[4], wf[2][4];

    // Output descriptors of the tile being flushed (base = first row of the tile, extent = its valid rows)
    struct OutDesc { pp_rsrc out, out2, side; int n0, m0; };
    auto out_desc = [&](int mt, int nt) -> OutDesc {
        OutDesc o;
        const int rows = M - mt * G::BM;
        // (GATHER: the rows scatter over the whole dIn tensor: one descriptor for all of it, offsets from its start)
        const unsigned bytes = GATHER == 1 ? (unsigned)((size_t)(M / (cv.Ho * cv.Wo)) * cv.H * cv.W * cv.Cin * 2)
                                           : (unsigned)((rows > G::BM ? G::BM : rows) * ep.ld * 2);
        const size_t base = GATHER == 1 ? 0 : (size_t)mt * G::BM * ep.ld;
        o.m0 = mt * G::BM;
        o.out = pp_make_rsrc(ep.out + base, (ABL & 32) ? 0u : bytes);          // (ABL 32: stores issued but dropped)
        o.out2 = pp_make_rsrc(ep.out2 ? ep.out2 + base : ep.out, ep.out2 ? bytes : 0u);
        o.side = pp_make_rsrc(ep.side ? ep.side + base : ep.out, ep.side ? bytes : 0u);
        o.n0 = nt * G::BN;
        return o;
    };
    // Flush of row block i (rows wr * 128 + i * 32 + [0, 32), the wave's 64 columns).  Accumulator layout: lane = (row l31, half hi),
    // registers = columns 32 j + 8 g + 4 hi + w.  Constants are added there (fp32), the rounded bf16 values go through the wave
    // pair's LDS scratch and come back as 16-byte pieces of FULL 128-byte rows: lane -> (row it * 8 + lane / 8, piece lane % 8),
    // so that a store instruction writes 8 complete rows.  Side inputs (residual / factor) are loaded and applied in that row form.
    char* const scr = smem + G::OFF_SCR + wc * G::SCR_BYTES;
    // global offset of the 16-byte piece (row it * 8 + lane / 8 of block i, piece lane % 8) of the tile being flushed
    auto row_voff = [&](int i, int it, const OutDesc& od, int lane) __attribute__((always_inline)) {
        const int r = wr * 128 + i * 32 + it * 8 + (lane >> 3), pc = lane & 7, colb = od.n0 + wc * 64;
        int voff = (r * ep.ld + colb + 8 * pc) * 2;
        if (GATHER == 1) {                            // row (frame, a, b), wave column (py, px, channel block) -> pixel (2a + py, 2b + px)
            const int m = od.m0 + r;
            uint32_t f, rem, a, b;
            cv.dHoWo.divmod((uint32_t)(m < M ? m : 0), f, rem);
            cv.dWo.divmod(rem, a, b);
            const int cls = colb / cv.Cin, cb = colb - cls * cv.Cin;
            const int pix = ((int)f * cv.H + 2 * (int)a + (cls >> 1)) * cv.W + 2 * (int)b + (cls & 1);
            voff = m < M ? (pix * cv.Cin + cb + 8 * pc) * 2 : (int)0xfffffff0u;
        }
        return voff;
    };
    // side rows (residual / factor / added cotangent) of block i, in the row form the stores use.  Issued ONE flushed block ahead
    // (step_body): vmcnt retires in order, so a load issued right before its use would wait for every LDS-DMA piece in flight.
    auto side_load = [&](int i, const OutDesc& od, u32x4 (&sd)[4]) __attribute__((always_inline)) {
        if (!SIDE || (ABL & 16)) return;
        int lane = lane_;
#ifndef RVT_EMU
        asm volatile("" : "+v"(lane));
#endif
#pragma unroll
        for (int it = 0; it < 4; it++) sd[it] = pp_load16(od.side, row_voff(i, it, od, lane));
    };
    auto flush_block = [&](int i, const OutDesc& od, const u32x4 (&sd)[4]) __attribute__((always_inline)) {
        if (ABL & 16) {                               // keep the accumulators alive (or the MFMAs are dead code)
#ifndef RVT_EMU
            asm volatile("" ::"v"(acc[i][0]), "v"(acc[i][1]));
#endif
            return;
        }
        // everything lane-derived below is recomputed per flush from an OPAQUE copy of the lane id: hoisted out of the step loop these
        // per-lane addresses cost registers the main loop does not have (spills there = vmcnt(0) drains of the load stream)
        int lane = lane_, l31, hi;
#ifndef RVT_EMU
        asm volatile("" : "+v"(lane));
#endif
        l31 = lane & 31; hi = lane >> 5;
        const int row0 = wr * 128 + i * 32;           // first tile row of the block
        const int colb = od.n0 + wc * 64;             // first column
        typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
        // block j of the wave: constants in the accumulator layout, rounded values -> scratch[row l31][32 j + 8 g + 4 hi + w]
        auto convert = [&](int j, bool second) __attribute__((always_inline)) {
            float v[16];
#pragma unroll
            for (int g = 0; g < 4; g++) {
                const int n = colb + 32 * j + 8 * g + 4 * hi;
                const f32x4 c0 = *reinterpret_cast<const f32x4*>(cst + n);
#pragma unroll
                for (int w = 0; w < 4; w++) v[4 * g + w] = acc[i][j][4 * g + w] + c0[w];
                if (EP == PP_SCALE_RES) {
                    const f32x4 c1 = *reinterpret_cast<const f32x4*>(cst + c1off + n);
#pragma unroll
                    for (int w = 0; w < 4; w++) v[4 * g + w] *= c1[w];
                }
            }
            if (EP == PP_GELU_DUAL) {
#pragma unroll
                for (int h8 = 0; h8 < 2; h8++) {
                    float x8[8], g8[8], p8[8];
#pragma unroll
                    for (int e = 0; e < 8; e++) x8[e] = v[8 * h8 + e];
                    gelu_both_8(x8, g8, p8);
#pragma unroll
                    for (int e = 0; e < 8; e++) v[8 * h8 + e] = second ? p8[e] : g8[e];
                }
            }
#pragma unroll
            for (int g = 0; g < 4; g++) {
                bf16x4 t;
#pragma unroll
                for (int w = 0; w < 4; w++) t[w] = (bf16)v[4 * g + w];
                *reinterpret_cast<bf16x4*>(scr + l31 * G::SCR_LD + (32 * j + 8 * g + 4 * hi) * 2) = t;
            }
        };
        auto store_rows = [&](const pp_rsrc& dst, bool apply_side) __attribute__((always_inline)) {
#pragma unroll
            for (int it = 0; it < 4; it++) {
                const int r = it * 8 + (lane >> 3), pc = lane & 7;
                const int voff = row_voff(i, it, od, lane);
                u32x4 t = *reinterpret_cast<const u32x4*>(scr + r * G::SCR_LD + pc * 16);
                if (SIDE && apply_side) {
                    bf16x8 a, b;
                    __builtin_memcpy(&a, &t, 16);
                    __builtin_memcpy(&b, &sd[it], 16);
                    float o[8];
#pragma unroll
                    for (int e = 0; e < 8; e++) o[e] = EP == PP_MUL ? (float)a[e] * (float)b[e] : (float)a[e] + (float)b[e];
                    const bf16x8 rv = frag_from_float<bf16>(o);
                    __builtin_memcpy(&t, &rv, 16);
                }
                pp_store16(dst, voff, t);
            }
        };
        // (LDS operations of one wave execute in order: no synchronisation between the scratch writes and the read-back;
        //  the emulator runs the lanes one after the other and needs the rendezvous)
        convert(0, false);
        convert(1, false);
        pp_wave_sync();
        store_rows(od.out, true);
        if (EP == PP_GELU_DUAL) {
            pp_wave_sync();
            convert(0, true);
            convert(1, true);
            pp_wave_sync();
            store_rows(od.out2, false);
        }
        pp_wave_sync();
    };

    // ---- prologue: units 0..11 (step 0 complete, W of step 1) ----
    __syncthreads();                                  // column constants in LDS; nothing else in flight yet
    Desc d1, d2;
    {
        const Desc d0 = next_desc();
#pragma unroll
        for (int g = 0; g < 4; g++) issue_w(d0, 0, g);
#pragma unroll
        for (int i = 0; i < 4; i++) issue_x(d0, 0, i);
        d1 = next_desc();
#pragma unroll
        for (int g = 0; g < 4; g++) issue_w(d1, 1, g);
        d2 = next_desc();
    }
    if (!(ABL & 1)) pp_wait_vm<7>();                  // W and X0 of step 0 have landed
    pp_barrier();
    if (wr == 1) pp_barrier();                        // group 1 runs one barrier behind group 0

    // One K-tile step = four phases; phase p = MFMA row block p of every wave (32 token rows x the wave's 64 weight rows).
    // Load-stream unit order per step: W0 W1 W2 W3 X0 X1 X2 X3 (index 8 step + u), two units per phase, phase k issues units
    // 2 k + 12 and 2 k + 13:   p0 -> X0 X1 of step s+1,  p1 -> X2 X3 of s+1,  p2 -> W0 W1 of s+2,  p3 -> W2 W3 of s+2.
    //   RAW  phase k waits for what phase k+1 reads (X block p+1; at p3 all of W and X0 of the next step): with nothing but the
    //        stream in flight the counts are 8, 9, 10, 7 instructions younger than the needed unit;
    //   WAR  a unit overwrites the one 16 indices earlier (two stages), last read >= 2 phases before the issue (W: read at p0 only).
    // MODE (compile time): 0 plain; 1 first K tile of a tile, the previous tile is flushed (row block p in phase p, E = EPI_OPS
    // more instructions per flushed block: they retire in order with the loads, so they are added to the later counts);
    // 2 the step after a MODE 1 step; 3 first K tile of the workgroup's first tile.
    // FIRST (modes 1, 3): the accumulators start from zero (C operand = 0 of the first k-step).
    int s = 0;
    OutDesc od = out_desc(0, 0);
    auto bar = [&]() { if (!(ABL & 8)) pp_barrier(); };
    auto xfrag = [&](int st, int i, int ks) __attribute__((always_inline)) {
        if (ABL & 2) { bf16x8 v; for (int e = 0; e < 8; e++) v[e] = (bf16)(float)(st + i + ks + lane); return v; }
        return *reinterpret_cast<const bf16x8*>(smem + ((st + xaddr0) ^ (ks << 5)) + i * G::UNIT);
    };
    auto wfrag = [&](int st, int j, int ks) __attribute__((always_inline)) {
        if (ABL & 2) { bf16x8 v; for (int e = 0; e < 8; e++) v[e] = (bf16)(float)(st + j + ks - lane); return v; }
        return *reinterpret_cast<const bf16x8*>(smem + ((st + waddr0) ^ (ks << 5)) + j * 4096);
    };
    auto step_body = [&](auto mode_c) __attribute__((always_inline)) {
        constexpr int MODE = decltype(mode_c)::value;
        constexpr bool FIRST = MODE == 1 || MODE == 3, FLUSH = MODE == 1;
        // vector memory instructions issued by the flush phases p0..p3 of a MODE 1 step: the block's stores, the side rows of the
        // NEXT block (p0: of blocks 0 and 1)
        constexpr int ST = (ABL & 16) ? 0 : (EP == PP_GELU_DUAL ? 8 : 4) + ((ABL & 64) ? 4 : 0);      // (ABL 64, timing probe only: counts too lax)
        constexpr int SL = (SIDE && !(ABL & 16)) ? 4 : 0;
        // (GATHER: the scatter addresses leave no registers for a second set of side rows: they are loaded right before their use)
        constexpr bool PREF = GATHER != 1;
        constexpr int E0 = ST + (PREF ? 2 * SL : SL), E1 = ST + SL, E2 = ST + SL, E3 = ST + (PREF ? 0 : SL);
        constexpr int W0 = 8 + (MODE == 2 ? E0 + E1 + E2 + E3 : 0), W1 = 9 + (MODE == 1 ? E0 : MODE == 2 ? E1 + E2 + E3 : 0),
                      W2 = 10 + (MODE == 1 ? E0 + E1 : MODE == 2 ? E1 + E2 + E3 : 0), W3 = 7 + (MODE == 1 ? E0 + E1 + E2 : 0);
        u32x4 sda[4], sdb[4];
        int st = (s & 1) * G::STAGE;
#ifndef RVT_EMU
        asm volatile("" : "+v"(st));                  // opaque: the fragment addresses are formed per step from two base registers (hoisted, they spill)
#endif
        auto mfmas = [&](int i) __attribute__((always_inline)) {
            pp_setprio(1);
#pragma unroll
            for (int ks = 0; ks < 4; ks++) {
                if (ABL & 4) { if (FIRST && ks == 0) { acc_zero(acc[i][0]); acc_zero(acc[i][1]); } acc[i][0][ks] += (float)wf[0][ks][0] * (float)xf[ks][1]; continue; }
                if (FIRST && ks == 0) { mma32_zero(acc[i][0], wf[0][ks], xf[ks]); mma32_zero(acc[i][1], wf[1][ks], xf[ks]); }
                else { mma32(acc[i][0], wf[0][ks], xf[ks]); mma32(acc[i][1], wf[1][ks], xf[ks]); }
            }
            pp_setprio(0);
        };
        // ---------------- phase 0 ----------------
        issue_x(d1, (s + 1) & 1, 0); issue_x(d1, (s + 1) & 1, 1);
        if (!(ABL & 1)) pp_wait_vm<W0>();
        if (FLUSH) { side_load(0, od, sda); if (PREF) side_load(1, od, sdb); flush_block(0, od, sda); }
#pragma unroll
        for (int ks = 0; ks < 4; ks++) { wf[0][ks] = wfrag(st, 0, ks); xf[ks] = xfrag(st, 0, ks); wf[1][ks] = wfrag(st, 1, ks); }
        bar();
        mfmas(0);
        bar();
        // ---------------- phase 1 ----------------
        issue_x(d1, (s + 1) & 1, 2); issue_x(d1, (s + 1) & 1, 3);
        if (!(ABL & 1)) pp_wait_vm<W1>();
        if (FLUSH) { if (PREF) { side_load(2, od, sda); flush_block(1, od, sdb); } else { side_load(1, od, sda); flush_block(1, od, sda); } }
#pragma unroll
        for (int ks = 0; ks < 4; ks++) xf[ks] = xfrag(st, 1, ks);
        bar();
        mfmas(1);
        bar();
        // ---------------- phase 2 ----------------
        issue_w(d2, s & 1, 0); issue_w(d2, s & 1, 1);
        if (!(ABL & 1)) pp_wait_vm<W2>();
        if (FLUSH) { if (PREF) { side_load(3, od, sdb); flush_block(2, od, sda); } else { side_load(2, od, sda); flush_block(2, od, sda); } }
#pragma unroll
        for (int ks = 0; ks < 4; ks++) xf[ks] = xfrag(st, 2, ks);
        bar();
        mfmas(2);
        bar();
        // ---------------- phase 3 ----------------
        issue_w(d2, s & 1, 2); issue_w(d2, s & 1, 3);
        if (!(ABL & 1)) pp_wait_vm<W3>();
        if (FLUSH) { if (PREF) flush_block(3, od, sdb); else { side_load(3, od, sda); flush_block(3, od, sda); } }
#pragma unroll
        for (int ks = 0; ks < 4; ks++) xf[ks] = xfrag(st, 3, ks);
        bar();
        mfmas(3);
        bar();
        d1 = d2;
        d2 = next_desc();
        s++;
    };
    for (int u = 0; u < work.count; u++) {
        int mt, nt;
        work.tile(u, mt, nt);
        const int nk = nk_of(nt);
        if (u == 0) {
            step_body(std::integral_constant<int, 3>());
            for (int kt = 1; kt < nk; kt++) step_body(std::integral_constant<int, 0>());
        } else {
            step_body(std::integral_constant<int, 1>());
            step_body(std::integral_constant<int, 2>());
            for (int kt = 2; kt < nk; kt++) step_body(std::integral_constant<int, 0>());
        }
        od = out_desc(mt, nt);
    }
    if (wr == 0) pp_barrier();                        // pairs with group 1's extra barrier
    pp_wait_vm<0>();                                  // (trailing empty-buffer pieces still write LDS)
    // last tile: the two waves of a pair share a scratch, so the groups flush one after the other
    if (wr == 0) {
#pragma unroll
        for (int i = 0; i < 4; i++) { u32x4 sd[4]; side_load(i, od, sd); flush_block(i, od, sd); }
    }
    pp_barrier();
    if (wr == 1) {
#pragma unroll
        for (int i = 0; i < 4; i++) { u32x4 sd[4]; side_load(i, od, sd); flush_block(i, od, sd); }
    }
}

// host side -------------------------------------------------------------------------------------------------------
inline bool ppgemm_shape_ok(int M, int N, int K, int ldx, int ldw, int kcut) {
    const int ldmax = ldx > ldw ? (ldx > N ? ldx : N) : (ldw > N ? ldw : N);
    return M >= 256 && N % 256 == 0 && N <= PPGeom::MAX_CST && K % 64 == 0 && K >= 128 && kcut % 64 == 0 && ldx % 8 == 0 &&
           ldw % 8 == 0 && (size_t)256 * (size_t)ldmax * 2 < (1ull << 31);
}

template <int EP, int GATHER = 0>
inline void launch_ppgemm(const PPMat& X, const PPMat& W, const PPEpArgs& ep, int M, int N, int K, hipStream_t stream,
                          const PPConv& cv = PPConv()) {
    const int m_tiles = (M + 255) / 256, n_tiles = N / 256;
    const int grid_override = g_tuning.ppgemm_grid;      // (tests: small grids walk several tiles)
    // panels are dealt to XCDs (PPWork): XCD 0 owns the most tiles, ceil(m_tiles / 8) * n_tiles; a grid cut to the TOTAL tile
    // count gives every XCD total / 8 workgroups and the fullest XCD's first workgroups a second tile - twice the critical path
    // on a one-round launch (5760 x 2048 x 1024: 184 tiles, 56 us at 184 workgroups).  Size the per-XCD share by the fullest XCD.
    const int per_xcd = ((m_tiles + 7) / 8) * n_tiles;
    int grid = grid_override > 0 ? grid_override : 256;
    if (grid > 8 * per_xcd) grid = 8 * per_xcd;
    grid = (grid + 7) & ~7;
    hipLaunchKernelGGL((ppgemm_kernel<EP, 0, GATHER>), dim3(grid), dim3(512), 0, stream, X, W, ep, M, N, K, m_tiles, n_tiles, cv);
}

}  // namespace rvt
