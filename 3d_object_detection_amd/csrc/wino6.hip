// Winograd F(4x4,3x3) for the stride-1 3x3 convolutions of the backbone (networks/pointpillars8_shared.py:114-181,418-431:
// Resnet2 = x + convs(IN -> ReLU -> conv3x3 ...)) on v_mfma_f32_16x16x4_f32 -- 36 MFMA positions per 4x4 output tile instead
// of 16 per 2x2 tile (wino4_mfma, conv.hip): 2.25 multiplies per output pixel and channel pair against 4 (direct: 9).
//
//   Y = A^T [ (G g G^T) (.) (B^T d B) ] A,   d = 6x6 input window of relu(norm(x)), g = 3x3 filter, Y = 4x4 outputs
//
// On gfx950 the fp32 MFMA runs at the fp32 VECTOR rate and nothing of a lone wave overlaps it (tools/issue_probe.hip), so a
// kernel's time is (MFMAs x 32 cycles) + (every other VALU instruction x 4) + epilogues: F(4x4) cuts the first term 1.78x
// and keeps the others per output pixel about where F(2x2) has them.
//
// * 36 positions x 64 output channels x 16 tiles = 576 accumulator registers per lane: more than a wave owns.  The POSITIONS
//   are split over the four waves of a workgroup (one wave per SIMD, 9 positions x 4 M-tiles x 4 = 144 accumulators each):
//       wave 0: row 0 of the 6x6 position grid + the left half (columns 0..2) of row 1      wave 1: right half of row 1 + row 2
//       wave 2: row 3 + left half of row 4                                                   wave 3: right half of row 4 + row 5
//   A wave then needs only TWO rows of B^T d (25-28 packed VALU instructions per 36 MFMAs), every wave multiplies the same 16
//   tiles (one 16x16-pixel patch, 18x18 with halo, staged once per workgroup) and all 64 output channels of the block.
// * The transformed weights (36 x Cin x 64 floats per block: 4x the 3x3 filter) are PRIVATE to a wave (its 9 positions), so
//   they never touch LDS: packed on the host as [block][k-step][wave][position][lane][M-tile] they are the A operands of a
//   k-step as they lie in memory, fetched global -> register (three AGPR buffers, one VGPR buffer) by 9 coalesced dwordx4 requests
//   per k-step, TWO chunks ahead (vmcnt returns in order: they queue behind the HBM patch pieces); L2-resident: 590 KB per 64->64 layer.
// * LDS holds the input patches only: a 3-deep ring of 8-channel chunks [c][row][RS] (normalised, ReLU'd, zero-padded while
//   staging: dwordx4 pieces, 32 threads per channel), walked by a load side that runs two chunks ahead of the MFMAs across
//   tile boundaries, as in wino4_mfma.
// * Epilogue: Y = A^T M A is linear in M, so every wave applies the COLUMN half of the output transform to its own positions
//   (T[i][x] = sum_j A^T[x][j] M[i][j], 4 values per full row, 4 per half row), the waves exchange those through LDS (96 KB,
//   one barrier) and wave w finishes M-tile w -- rows 16 w .. 16 w + 15 of the block for all 256 pixels: row half of the
//   transform, residual add, dwordx4 row stores, InstanceNorm statistics (no cross-wave reduction: a wave owns its rows).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include "conv_common.h"

namespace ppc {
namespace {

typedef float f32x2 __attribute__((ext_vector_type(2)));

#ifndef PP_W6_DIAG
#define PP_W6_DIAG 0 // timing-only ablations (wrong results): 1 no input transform, 2 no raw LDS reads, 4 no A loads, 8 no MFMA, 16 no staging, 32 no epilogue exchange, 64 load side frozen (patch loads hit L2), 128 A loads always the same 72 KB, 2048 every A load lands in the VGPR buffer (is the AGPR destination what costs?), 4096 A loads of 4 bytes per lane (requests or bytes?), 16384 no epilogue
#endif

template <int TWT>
struct Wino6Cfg {
    static constexpr int THT = 16 / TWT;             // the 16 tiles of a workgroup: TWT x THT tiles of 4x4 pixels
    static constexpr int PW = 4 * TWT, PH = 4 * THT; // output patch
    static constexpr int IW = PW + 2, IH = PH + 2;   // input patch with halo
    // LDS row stride (floats): a multiple of 4 (dwordx4 pieces, ds_read_b128) with the 16 lanes of a ds_read_b128 group on
    // 16 distinct bank quads: start / 4 = (4 ty + a) RS / 4 + tx  ->  TWT = 4 wants RS = 4 or 12 (mod 16)
    static constexpr int rs()
    {
        int v = (IW + 3) & ~3;
        if (TWT == 4) while (v % 16 != 4 && v % 16 != 12) v += 4;
        return v;
    }
    static constexpr int RS = rs();
    static constexpr int NQ = RS / 4; // dwordx4 pieces per patch row
    // TWT = 1 (the 4-pixel column strips: 16 tiles stacked): a row stride that is a multiple of 4 floats puts the 16 tiles' rows on
    // at most 4 bank quads (8-way conflicts with RS = 8: 30x the conflict cycles of the other families, profiles/r04_pmc_wino.txt).
    // Patch row r therefore starts at r RS + SK (r / 4): tile ty's rows begin at quad 9 ty + ... -- 16 distinct quads.
    static constexpr int SK = TWT == 1 ? 4 : 0;
    static constexpr int row_at(int r) { return r * RS + SK * (r / 4); }
    static constexpr int cs()         // channel stride: a multiple of 64 floats (the kq halves of a read group stay apart)
    {
        int v = row_at(IH - 1) + RS;
        while (v % (TWT == 1 ? 32 : 64)) ++v; // (the skewed strip patch at a multiple of 64 would not fit the 160 KB next to the exchange buffer)
        return v;
    }
    static constexpr int CS = cs();
    static constexpr int KC = 8;      // channels per chunk = 2 k-steps of 4
    static constexpr int NSTAGE = 3;
    static constexpr int LDS_IN = KC * CS;
    static constexpr int NPC = IH * NQ;           // pieces per channel
    static constexpr int PRND = (NPC + 31) / 32;  // pieces per thread and chunk (32 threads stage one channel)
    static constexpr int XB = 4 * 3 * 2 * 4 * 64 * 4; // exchange: [dst M-tile][src][full | half][row r][lane][x] floats
    static constexpr int LDS_FLOATS = NSTAGE * LDS_IN + XB + 2 * 640;
    static constexpr int THREADS = 256;
    static constexpr int BM = 64;
    static_assert(LDS_FLOATS * 4 <= 160 * 1024, "must fit the 160 KB LDS");
    static_assert(PRND <= 8, "staging schedule: at most two pieces per VALU cluster");
    // ---- the vector-memory operations of one chunk, in issue order (they are ALL issued from asm statements, so the kernel counts
    // them itself: vmcnt returns in order, one counter for everything).  Per k-step ks and position j: the A request of (ks, j)
    // right behind its MFMAs; behind positions 5 and 7 the requests of the staging pieces written there.
    static constexpr int piece_at(int ks, int j, int w) { return j == 5 ? (w == 0 ? 2 * ks : 2 * ks + 4) : j == 7 ? (w == 0 ? 2 * ks + 1 : 2 * ks + 5) : 99; }
    static constexpr int idx_A(int ks, int j)
    {
        int n = 0;
        for (int k = 0; k < 2; ++k)
            for (int jj = 0; jj < 9; ++jj) {
                if (k == ks && jj == j) return n;
                ++n;
                for (int w = 0; w < 2; ++w) if (piece_at(k, jj, w) < PRND) ++n;
            }
        return n;
    }
    static constexpr int idx_P(int pc)
    {
        int n = 0;
        for (int k = 0; k < 2; ++k)
            for (int jj = 0; jj < 9; ++jj) {
                ++n;
                for (int w = 0; w < 2; ++w) if (piece_at(k, jj, w) < PRND) { if (piece_at(k, jj, w) == pc) return n; ++n; }
            }
        return n;
    }
    static constexpr int T = 18 + PRND;          // operations per chunk
    static constexpr int N_A = 2 * T - 1;        // operations younger than an A request when its MFMAs are due two chunks later
    // piece pc is requested behind position j_p of k-step ks_p and normalised two chunks later in the gap of position j_p - 1 (behind that
    // position's A request): operations younger than its request at that point
    // (capped at the counter's 63: a smaller count only waits for a few more of the oldest requests)
    static constexpr int n_piece(int pc) { const int ks = (pc & 3) >> 1, j = (pc & 1) ? 7 : 5; const int n = (T - 1 - idx_P(pc)) + T + idx_A(ks, j - 1) + 1; return n > 63 ? 63 : n; }
    static_assert(N_A <= 63, "vmcnt is a 6-bit counter");
};

#if PP_W6_DIAG & 256
#define W6_SYNC "\n\ts_waitcnt vmcnt(0)" // debugging: every request completes before the next instruction (tests the logic without anything in flight)
#else
#define W6_SYNC ""
#endif
// hipcc pads no hazard inside an asm statement: an SGPR operand it has just reloaded from a spill lane (v_readlane_b32, a VALU write
// of an SGPR) needs 5 wait states before a VMEM instruction reads it as descriptor or offset
#define W6_SGPR_PAD "s_nop 4\n\t"
typedef int i32x4 __attribute__((ext_vector_type(4)));
// raw buffer descriptor (stride 0, dword format) as four scalars for the asm loads
__device__ __forceinline__ i32x4 w6_rsrc(const void* ptr, unsigned bytes)
{
    const unsigned long long a = (unsigned long long)ptr;
    return (i32x4){(int)(unsigned)a, (int)((unsigned)(a >> 32) & 0xFFFFu), (int)bytes, 0x00020000};
}

// (x.lo + y.hi, x.lo - y.hi)
__device__ __forceinline__ f32x2 pk_lo_pm_hi(f32x2 x, f32x2 y)
{
    f32x2 r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(x), "v"(y));
    return r;
}
// (w.lo + 2 z.hi, w.lo - 2 z.hi); two = (2, 2)
__device__ __forceinline__ f32x2 pk_lo_pm2_hi(f32x2 w, f32x2 z, f32x2 two)
{
    f32x2 r;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0] neg_hi:[0,1,0]" : "=v"(r) : "v"(z), "v"(two), "v"(w));
    return r;
}
__device__ __forceinline__ f32x2 pkfma(f32x2 a, float k, f32x2 c) { return __builtin_elementwise_fma(a, (f32x2){k, k}, c); }
__device__ __forceinline__ f32x2 lo2(f32x4 v) { return __builtin_shufflevector(v, v, 0, 1); }
__device__ __forceinline__ f32x2 hi2(f32x4 v) { return __builtin_shufflevector(v, v, 2, 3); }

// One 1-D pass of B^T over six values held as pairs P0 = (v0, v1), P1 = (v2, v3), P2 = (v4, v5):
//   c0 = 4 v0 - 5 v2 + v4          c1, c2 = (v4 - 4 v2) +- (v3 - 4 v1)          c3, c4 = (v4 - v2) +- 2 (v3 - v1)          c5 = 4 v1 - 5 v3 + v5
// PART: 0 all six, 1 the left half (c0, c1, c2), 2 the right half (c3, c4, c5)
template <int PART>
__device__ __forceinline__ void w6_bt_pairs(const f32x2 P0, const f32x2 P1, const f32x2 P2, const f32x2 two, float* c)
{
    if constexpr (PART == 0) {
        const f32x2 e = pkfma(P1, -5.f, pkfma(P0, 4.f, P2));
        c[0] = e[0]; c[5] = e[1];
    } else if constexpr (PART == 1) {
        c[0] = __builtin_fmaf(P1[0], -5.f, __builtin_fmaf(P0[0], 4.f, P2[0]));
    } else {
        c[2] = __builtin_fmaf(P1[1], -5.f, __builtin_fmaf(P0[1], 4.f, P2[1]));
    }
    if constexpr (PART != 2) {
        const f32x2 u = pkfma(P1, -4.f, P2), v = pkfma(P0, -4.f, P1);
        const f32x2 x = pk_lo_pm_hi(u, v);
        c[1] = x[0]; c[2] = x[1];
    }
    if constexpr (PART != 1) {
        const f32x2 w = P2 - P1, z = P1 - P0;
        const f32x2 y = pk_lo_pm2_hi(w, z, two);
        if constexpr (PART == 0) { c[3] = y[0]; c[4] = y[1]; } else { c[0] = y[0]; c[1] = y[1]; }
    }
}

// position (i, jj) of the 6x6 grid owned by wave WV as its local position j = 0..8: j < 6 the full row, j >= 6 the half row
__host__ __device__ constexpr int w6_pos_i(int wv, int j) { return j < 6 ? (wv == 0 ? 0 : wv == 1 ? 2 : wv == 2 ? 3 : 5) : (wv < 2 ? 1 : 4); }
__host__ __device__ constexpr int w6_pos_j(int wv, int j) { return j < 6 ? j : (j - 6) + ((wv & 1) ? 3 : 0); }

// ---- the accumulator file is laid out by hand (as in wino4_mfma: handed 144 accumulators plus 144 operand registers next to the
// transforms, hipcc spills):  a[0:143]   accumulators, quad (position j, M-tile mt) at (j*4 + mt)*4
//                             a[144:251] A operands of k-step buffers 0..2, (buffer kb, position j) at 144 + (kb*9 + j)*4 (4 M-tiles)
// The fourth A buffer lives in VGPRs.  Every asm statement that names AGPRs clobbers all of them, so hipcc allocates the whole
// accumulator half and never parks a value there (audit: tools/isa_stats.py -- no v_accvgpr_* outside asm, no scratch).
#define W6_A10(b) "a" #b "0", "a" #b "1", "a" #b "2", "a" #b "3", "a" #b "4", "a" #b "5", "a" #b "6", "a" #b "7", "a" #b "8", "a" #b "9"
#define W6_A100(h) W6_A10(h##0), W6_A10(h##1), W6_A10(h##2), W6_A10(h##3), W6_A10(h##4), W6_A10(h##5), W6_A10(h##6), W6_A10(h##7), W6_A10(h##8), W6_A10(h##9)
#define W6_AGPRS "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", W6_A10(1), W6_A10(2), W6_A10(3), W6_A10(4), W6_A10(5), W6_A10(6), W6_A10(7), W6_A10(8), W6_A10(9), \
                 W6_A100(1), W6_A10(20), W6_A10(21), W6_A10(22), W6_A10(23), W6_A10(24), "a250", "a251", "a252", "a253", "a254", "a255"

// the four MFMAs (M-tiles 0..3) of position J with the A operands of k-step buffer KB; ZERO: C = 0 (a tile's first k-step)
template <int J, int KB, bool ZERO>
__device__ __forceinline__ void w6_mfma4(const float b, const f32x4& a3)
{
    constexpr int c0 = J * 16, a0 = 144 + (KB * 9 + J) * 4;
    if constexpr (PP_W6_DIAG & 8) { asm volatile("" ::"v"(b)); return; }
    if constexpr (KB < 3) {
        if constexpr (ZERO)
            asm volatile("v_mfma_f32_16x16x4_f32 a[%c1:%c2], a[%c9], %0, 0\n\tv_mfma_f32_16x16x4_f32 a[%c3:%c4], a[%c10], %0, 0\n\t"
                         "v_mfma_f32_16x16x4_f32 a[%c5:%c6], a[%c11], %0, 0\n\tv_mfma_f32_16x16x4_f32 a[%c7:%c8], a[%c12], %0, 0"
                         :: "v"(b), "i"(c0), "i"(c0 + 3), "i"(c0 + 4), "i"(c0 + 7), "i"(c0 + 8), "i"(c0 + 11), "i"(c0 + 12), "i"(c0 + 15),
                            "i"(a0), "i"(a0 + 1), "i"(a0 + 2), "i"(a0 + 3) : W6_AGPRS);
        else
            asm volatile("v_mfma_f32_16x16x4_f32 a[%c1:%c2], a[%c9], %0, a[%c1:%c2]\n\tv_mfma_f32_16x16x4_f32 a[%c3:%c4], a[%c10], %0, a[%c3:%c4]\n\t"
                         "v_mfma_f32_16x16x4_f32 a[%c5:%c6], a[%c11], %0, a[%c5:%c6]\n\tv_mfma_f32_16x16x4_f32 a[%c7:%c8], a[%c12], %0, a[%c7:%c8]"
                         :: "v"(b), "i"(c0), "i"(c0 + 3), "i"(c0 + 4), "i"(c0 + 7), "i"(c0 + 8), "i"(c0 + 11), "i"(c0 + 12), "i"(c0 + 15),
                            "i"(a0), "i"(a0 + 1), "i"(a0 + 2), "i"(a0 + 3) : W6_AGPRS);
    } else {
        static_assert(!(KB == 3 && ZERO), "the first k-step of a tile uses buffer 0");
        asm volatile("v_mfma_f32_16x16x4_f32 a[%c5:%c6], %1, %0, a[%c5:%c6]\n\tv_mfma_f32_16x16x4_f32 a[%c7:%c8], %2, %0, a[%c7:%c8]\n\t"
                     "v_mfma_f32_16x16x4_f32 a[%c9:%c10], %3, %0, a[%c9:%c10]\n\tv_mfma_f32_16x16x4_f32 a[%c11:%c12], %4, %0, a[%c11:%c12]"
                     :: "v"(b), "v"(a3[0]), "v"(a3[1]), "v"(a3[2]), "v"(a3[3]),
                        "i"(c0), "i"(c0 + 3), "i"(c0 + 4), "i"(c0 + 7), "i"(c0 + 8), "i"(c0 + 11), "i"(c0 + 12), "i"(c0 + 15) : W6_AGPRS);
    }
}
// the same four MFMAs of an AGPR buffer (KB < 3) with the NEXT request for that buffer's quad behind them in ONE statement: the request reads
// its SGPR operands at least five wait states (four MFMAs + s_nop 1) after anything hipcc may have put in front of the statement (a spill-lane
// reload, see W6_SGPR_PAD), so it needs no s_nop 4 of its own -- which costs a lone wave 12 cycles per request (tools/vmem_probe.hip: 1180 ->
// 1292 cycles per k-step of nine requests), 4 % of a chunk
template <int J, int KB, bool ZERO>
__device__ __forceinline__ void w6_mfma4_load(const float b, const i32x4 rw, const unsigned voff, const unsigned soff)
{
    static_assert(KB < 3, "buffer 3 lives in VGPRs");
    constexpr int c0 = J * 16, a0 = 144 + (KB * 9 + J) * 4;
    if constexpr (ZERO)
        asm volatile("v_mfma_f32_16x16x4_f32 a[%c1:%c2], a[%c9], %0, 0\n\tv_mfma_f32_16x16x4_f32 a[%c3:%c4], a[%c10], %0, 0\n\t"
                     "v_mfma_f32_16x16x4_f32 a[%c5:%c6], a[%c11], %0, 0\n\tv_mfma_f32_16x16x4_f32 a[%c7:%c8], a[%c12], %0, 0\n\t"
                     "s_nop 1\n\tbuffer_load_dwordx4 a[%c9:%c12], %13, %14, %15 offen" W6_SYNC
                     :: "v"(b), "i"(c0), "i"(c0 + 3), "i"(c0 + 4), "i"(c0 + 7), "i"(c0 + 8), "i"(c0 + 11), "i"(c0 + 12), "i"(c0 + 15),
                        "i"(a0), "i"(a0 + 1), "i"(a0 + 2), "i"(a0 + 3), "v"(voff), "s"(rw), "s"(soff) : W6_AGPRS);
    else
        asm volatile("v_mfma_f32_16x16x4_f32 a[%c1:%c2], a[%c9], %0, a[%c1:%c2]\n\tv_mfma_f32_16x16x4_f32 a[%c3:%c4], a[%c10], %0, a[%c3:%c4]\n\t"
                     "v_mfma_f32_16x16x4_f32 a[%c5:%c6], a[%c11], %0, a[%c5:%c6]\n\tv_mfma_f32_16x16x4_f32 a[%c7:%c8], a[%c12], %0, a[%c7:%c8]\n\t"
                     "s_nop 1\n\tbuffer_load_dwordx4 a[%c9:%c12], %13, %14, %15 offen" W6_SYNC
                     :: "v"(b), "i"(c0), "i"(c0 + 3), "i"(c0 + 4), "i"(c0 + 7), "i"(c0 + 8), "i"(c0 + 11), "i"(c0 + 12), "i"(c0 + 15),
                        "i"(a0), "i"(a0 + 1), "i"(a0 + 2), "i"(a0 + 3), "v"(voff), "s"(rw), "s"(soff) : W6_AGPRS);
}
// request the A operands (4 M-tiles: 16 bytes per lane) of (buffer KB, position J): voff = the lane's byte offset, soff = the (k-step, position)'s
template <int J, int KB>
__device__ __forceinline__ void w6_load_A(const i32x4 rw, const unsigned voff, const unsigned soff, f32x4& a3)
{
    if constexpr (PP_W6_DIAG & 4) return;
    if constexpr (PP_W6_DIAG & 4096) { // 4 bytes per lane instead of 16: the same requests with a quarter of the data
        constexpr int a0 = 144 + ((KB < 3 ? KB : 0) * 9 + J) * 4;
        asm volatile(W6_SGPR_PAD "buffer_load_dword a[%c3], %0, %1, %2 offen" :: "v"(voff), "s"(rw), "s"(soff), "i"(a0) : W6_AGPRS);
    } else if constexpr (KB < 3 && !(PP_W6_DIAG & 2048)) {
        constexpr int a0 = 144 + (KB * 9 + J) * 4;
        asm volatile(W6_SGPR_PAD "buffer_load_dwordx4 a[%c3:%c4], %0, %1, %2 offen" W6_SYNC :: "v"(voff), "s"(rw), "s"(soff), "i"(a0), "i"(a0 + 3) : W6_AGPRS);
    } else {
        asm volatile(W6_SGPR_PAD "buffer_load_dwordx4 %0, %1, %2, %3 offen" W6_SYNC : "=v"(a3) : "v"(voff), "s"(rw), "s"(soff));
    }
}
__device__ __forceinline__ void w6_load_x4(f32x4& dst, const i32x4 rs, const unsigned voff, const unsigned soff)
{
    if constexpr (PP_W6_DIAG & 16) { asm volatile("" : "=v"(dst)); return; }
    asm volatile(W6_SGPR_PAD "buffer_load_dwordx4 %0, %1, %2, %3 offen" W6_SYNC : "=v"(dst) : "v"(voff), "s"(rs), "s"(soff));
}
#if PP_W6_DIAG & 512
#define W6_STAMP(V) { __builtin_amdgcn_sched_barrier(0); V = __builtin_amdgcn_s_memtime(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_sched_barrier(0); }
#else
#define W6_STAMP(V)
#endif
template <int N> __device__ __forceinline__ void w6_wait() { asm volatile("s_waitcnt vmcnt(%c0)" :: "i"(N)); }
template <int N> __device__ __forceinline__ void w6_wait(f32x4& x) { asm volatile("s_waitcnt vmcnt(%c1)" : "+v"(x) : "i"(N)); }
// the 9 positions' accumulator rows (2 RP, 2 RP + 1) of M-tile MT as pairs, for v_pk_* arithmetic
template <int MT, int RP>
__device__ __forceinline__ void w6_acc_read(f32x2 (&mm)[9])
{
    float l[9], u[9];
    asm volatile("v_accvgpr_read_b32 %0, a[%c18+0]\n\tv_accvgpr_read_b32 %9, a[%c18+1]\n\t"
                 "v_accvgpr_read_b32 %1, a[%c18+16]\n\tv_accvgpr_read_b32 %10, a[%c18+17]\n\t"
                 "v_accvgpr_read_b32 %2, a[%c18+32]\n\tv_accvgpr_read_b32 %11, a[%c18+33]\n\t"
                 "v_accvgpr_read_b32 %3, a[%c18+48]\n\tv_accvgpr_read_b32 %12, a[%c18+49]\n\t"
                 "v_accvgpr_read_b32 %4, a[%c18+64]\n\tv_accvgpr_read_b32 %13, a[%c18+65]\n\t"
                 "v_accvgpr_read_b32 %5, a[%c18+80]\n\tv_accvgpr_read_b32 %14, a[%c18+81]\n\t"
                 "v_accvgpr_read_b32 %6, a[%c18+96]\n\tv_accvgpr_read_b32 %15, a[%c18+97]\n\t"
                 "v_accvgpr_read_b32 %7, a[%c18+112]\n\tv_accvgpr_read_b32 %16, a[%c18+113]\n\t"
                 "v_accvgpr_read_b32 %8, a[%c18+128]\n\tv_accvgpr_read_b32 %17, a[%c18+129]"
                 : "=v"(l[0]), "=v"(l[1]), "=v"(l[2]), "=v"(l[3]), "=v"(l[4]), "=v"(l[5]), "=v"(l[6]), "=v"(l[7]), "=v"(l[8]),
                   "=v"(u[0]), "=v"(u[1]), "=v"(u[2]), "=v"(u[3]), "=v"(u[4]), "=v"(u[5]), "=v"(u[6]), "=v"(u[7]), "=v"(u[8])
                 : "i"(MT * 4 + RP * 2)
                 : W6_AGPRS);
#pragma unroll
    for (int j = 0; j < 9; ++j) { mm[j][0] = l[j]; mm[j][1] = u[j]; }
}

template <int TWT, int WV, int ROOF>
__device__ __forceinline__ void wino6_body(const ConvP& p, float* smem)
{
    using C = Wino6Cfg<TWT>;
    constexpr int KC = C::KC, PRND = C::PRND;
    float* il = smem;                                                    // [3][KC][CS]
    float* xb = smem + C::NSTAGE * C::LDS_IN;                            // exchange buffer of the epilogue
    f32x2* aff = reinterpret_cast<f32x2*>(xb + C::XB);                   // [2 frame parities][320] (scale, shift)
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int m = lane & 15, kq = lane >> 4;
    const int chl = tid >> 5, l32 = tid & 31; // staging: channel of the chunk, piece lane

    const int nbx = p.rnbx, nby = p.rnby;
    const int ntile = nbx * nby, ncb = p.Cout / C::BM;
    const int total = ntile * ncb * p.nb;
    const int per = (total + 7) >> 3;
    const int xk = blockIdx.x & 7, xj = blockIdx.x >> 3, nloc = gridDim.x >> 3;
    const int lin_end = min(total, (xk + 1) * per);
    const int lin0 = xk * per + xj;
    if (lin0 >= lin_end) return;
    // Scalars of the per-tile address arithmetic, made opaque once: under SGPR pressure hipcc otherwise RE-LOADS kernel arguments from the
    // kernarg segment in the middle of the loop (s_load + s_waitcnt lgkmcnt(0): a scalar-cache miss costs microseconds) instead of keeping them
    int Hin = p.Hin, Win = p.Win, rx0 = p.rx0, ry0 = p.ry0, Cin = p.Cin;
    const float* in_base = p.in;
    size_t in_fs = p.in_fs;
    asm volatile("" : "+s"(Hin), "+s"(Win), "+s"(rx0), "+s"(ry0), "+s"(Cin), "+s"(in_base), "+s"(in_fs));
    const int nchunk = Cin / KC;
    const unsigned plane_b = (unsigned)(Hin * Win) * 4u;
    const unsigned chunk_wb = 2u * 4u * 9u * 1024u;               // bytes of transformed weights per chunk and block: 2 k-steps x 4 waves x 9 KB
    const unsigned block_wb = (unsigned)nchunk * chunk_wb;

    // ---------------- load side (input patches): walks the (item, chunk) stream FOUR chunks ahead of the compute side: two chunks in
    // the LDS ring, two in the registers -- vmcnt returns in order, so every load of the wave (the L2-resident weights included) must
    // be requested further ahead than an HBM load takes under load (2-3 us, more than one chunk of MFMAs) ----------------
    unsigned goff[PRND];
    int loff[PRND];
    unsigned vmask = 0u;
    i32x4 rin = w6_rsrc(in_base - 16, 0x7FFFFFFFu);
    const i32x4 rw = w6_rsrc(p.w, 0x7FFFFFFFu);
    f32x4 xv[2][PRND];
#pragma unroll
    for (int rd = 0; rd < PRND; ++rd) { // RS = 4 NQ: piece e sits at float 4 e (+ the skew of its row)
        const int e = min(l32 + 32 * rd, C::NPC - 1);
        loff[rd] = chl * C::CS + 4 * e + C::SK * ((e / C::NQ) / 4);
    }
    int s_lin = lin0, s_ch = 0, s_frame = 0, s_tab = 0;
    int r_c0[2] = {0, 0}, r_tab[2] = {0, 0}, r_lin[2] = {lin0, lin0}, mk_lin = -1; // the chunk held by register set 0 / 1
    unsigned r_vmask[2] = {0u, 0u};
    auto set_load_tile = [&](int l) {
        const int t_ = (l / ncb) % ntile, f_ = l / (ncb * ntile);
        const int iy0_ = ry0 + (t_ / nbx) * C::PH - 1, ix0_ = rx0 + (t_ % nbx) * C::PW - 1;
        vmask = 0u;
#pragma unroll
        for (int rd = 0; rd < PRND; ++rd) {
            const int e = min(l32 + 32 * rd, C::NPC - 1); // tail threads duplicate the last piece
            const int row = e / C::NQ, q = e - row * C::NQ;
            const int gy = iy0_ + row, gx = ix0_ + 4 * q;
            const bool rowok = gy >= 0 && gy < Hin;
            // a piece starts at most one float before its row (gx >= -1): the tensors wino6 reads carry a front pad, the descriptor
            // starts 16 floats inside it and the offsets carry those 64 bytes (a negative offset would read as out of range: zeros)
            goff[rd] = 64u + (unsigned)chl * plane_b + (rowok ? (unsigned)((gy * Win + gx) * 4) : 0u);
#pragma unroll
            for (int k = 0; k < 4; ++k) vmask |= ((rowok && gx + k >= 0 && gx + k < Win) ? 1u : 0u) << (rd * 4 + k);
        }
        rin = w6_rsrc(in_base + (size_t)f_ * in_fs - 16, 0x7FFFFFFFu);
        s_frame = f_;
    };
    auto load_aff = [&](int f_) {
        f32x2* dst = aff + s_tab * 320;
        for (int c = tid; c < Cin; c += C::THREADS)
            dst[c] = (f32x2){p.pre_scale[(size_t)f_ * p.aff_fs + c], p.pre_shift[(size_t)f_ * p.aff_fs + c]};
    };
    auto advance = [&]() {
        if constexpr (PP_W6_DIAG & 64) return;
        if (s_ch + 1 < nchunk) ++s_ch;
        else if (s_lin + nloc < lin_end) {
            const int f_old = s_frame;
            s_lin += nloc; s_ch = 0; set_load_tile(s_lin);
            if (s_frame != f_old) { s_tab ^= 1; load_aff(s_frame); }
        }
    };
    float mk[PRND][4]; // upper clamp of the normalised values of the pieces being written: +inf inside the image (= ReLU), 0 on the zero padding
    auto expand_mask = [&](unsigned vm) {
#pragma unroll
        for (int rd = 0; rd < PRND; ++rd)
#pragma unroll
            for (int k = 0; k < 4; ++k) mk[rd][k] = ((vm >> (rd * 4 + k)) & 1u) ? __builtin_inff() : 0.f;
    };
    f32x2 ss = {1.f, 0.f}; // (scale, shift) of this thread's channel of the chunk being written
#define W6_LOAD_PIECE(X, RD)                                                                     \
    { if constexpr ((RD) < PRND) w6_load_x4(xv[X][RD], rin, goff[RD], (unsigned)(s_ch * KC) * plane_b); }
#define W6_NORM_PIECE(X, RD, N)                                                                  \
    { if constexpr ((RD) < PRND && !(PP_W6_DIAG & 16)) {                                        \
        w6_wait<N>(xv[X][RD]);                                                                   \
        const f32x2 a_ = __builtin_elementwise_fma(lo2(xv[X][RD]), (f32x2){ss[0], ss[0]}, (f32x2){ss[1], ss[1]}); \
        const f32x2 b_ = __builtin_elementwise_fma(hi2(xv[X][RD]), (f32x2){ss[0], ss[0]}, (f32x2){ss[1], ss[1]}); \
        xv[X][RD] = (f32x4){__builtin_amdgcn_fmed3f(a_[0], 0.f, mk[RD][0]), __builtin_amdgcn_fmed3f(a_[1], 0.f, mk[RD][1]), \
                            __builtin_amdgcn_fmed3f(b_[0], 0.f, mk[RD][2]), __builtin_amdgcn_fmed3f(b_[1], 0.f, mk[RD][3])}; } }
#define W6_WRITE_PIECE(X, RD, IB)                                                                \
    { if constexpr ((RD) < PRND && !(PP_W6_DIAG & 16)) *reinterpret_cast<f32x4*>((IB) + loff[RD]) = xv[X][RD]; }

    // ---------------- compute side ----------------
    const int tx = m % TWT, ty = m / TWT;
    const int rbase = kq * C::CS + (4 * ty) * C::RS + C::SK * ty + 4 * tx;
    constexpr int A0 = (WV == 0) ? 0 : 1, A1 = (WV == 3) ? 5 : 4; // raw patch rows this wave's two rows of B^T d need
    const unsigned wlane = (unsigned)lane * 16u;
    const f32x2 two = {2.f, 2.f};

    f32x4 A3[9];     // A operands of k-step buffer 3 (the other three buffers are a[144:251])
    float B[2][9];   // B operands of the even / odd k-step
    f32x2 d[6][3];   // raw 6x6 window of one (channel, tile) as column pairs

    auto read_raw = [&](const float* base, auto R0, auto R1) {
        if constexpr (!(PP_W6_DIAG & 2)) {
            pp_steps<decltype(R0)::value, decltype(R1)::value>([&](auto AA) {
                constexpr int a = decltype(AA)::value;
                if constexpr (a >= A0 && a <= A1) {
                    constexpr int ro = a * C::RS + (a >= 4 ? C::SK : 0); // window row a of tile ty = patch row 4 ty + a
                    const f32x4 q = *reinterpret_cast<const f32x4*>(base + ro);
                    d[a][0] = lo2(q); d[a][1] = hi2(q);
                    d[a][2] = *reinterpret_cast<const f32x2*>(base + ro + 4);
                }
            });
        }
    };
    f32x2 F[3], H[3]; // the wave's full row and the source row of its half row of B^T d (over the window's columns, as pairs)
    auto row_pass = [&]() {
        if constexpr (!(PP_W6_DIAG & 1)) {
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                if constexpr (WV == 0) {
                    F[q] = pkfma(d[2][q], -5.f, pkfma(d[0][q], 4.f, d[4][q]));
                    H[q] = pkfma(d[2][q], -4.f, d[4][q]) + pkfma(d[1][q], -4.f, d[3][q]);
                } else if constexpr (WV == 1) {
                    const f32x2 u = pkfma(d[2][q], -4.f, d[4][q]), v = pkfma(d[1][q], -4.f, d[3][q]);
                    H[q] = u + v; F[q] = u - v;
                } else if constexpr (WV == 2) {
                    const f32x2 u = d[4][q] - d[2][q], v = d[3][q] - d[1][q];
                    F[q] = pkfma(v, 2.f, u); H[q] = pkfma(v, -2.f, u);
                } else {
                    H[q] = pkfma(d[3][q] - d[1][q], -2.f, d[4][q] - d[2][q]);
                    F[q] = pkfma(d[3][q], -5.f, pkfma(d[1][q], 4.f, d[5][q]));
                }
            }
        }
    };
    auto col_pass = [&](float* bn) {
        if constexpr (!(PP_W6_DIAG & 1)) {
            w6_bt_pairs<0>(F[0], F[1], F[2], two, bn);
            w6_bt_pairs<(WV & 1) ? 2 : 1>(H[0], H[1], H[2], two, bn + 6);
        } else {
#pragma unroll
            for (int j = 0; j < 9; ++j) bn[j] = d[A0 + (j % 4)][j % 3][j & 1];
        }
    };
    // weights of (block, chunk, k-step, position): byte offset inside the image (wave WV's slice)
    auto w_off = [&](unsigned chunk_base, int ks, int j) { return chunk_base + (unsigned)(ks * 36864 + WV * 9216 + j * 1024); };

    // ---------------- pipeline prologue: chunks 0 and 1 into ring slots 0 and 1, chunks 2 and 3 into the two register sets,
    // the A operands of chunks 0 and 1 into the four buffers ----------------
    set_load_tile(lin0);
    if (p.pre == PRE_STATS) {
        // one frame per launch (launch_conv hands the raw statistics over only then): the producer's fp64 sums become (scale, shift) here,
        // with norm_finalize's formula, instead of a launch of their own in front of the layer.  No frame change follows.
        const double* pa = p.pre_acc + (size_t)s_frame * p.pre_fs;
        for (int c = tid; c < Cin; c += C::THREADS) {
            double sm = 0.0, sq = 0.0;
#pragma unroll
            for (int r = 0; r < NREP; ++r) { sm += pa[((size_t)r * Cin + c) * 2]; sq += pa[((size_t)r * Cin + c) * 2 + 1]; }
            const double mean = sm * p.pre_inv_n;
            double var = sq * p.pre_inv_n - mean * mean;
            var = var > 0.0 ? var : 0.0;
            const double rstd = 1.0 / sqrt(var + (double)p.eps);
            aff[s_tab * 320 + c] = (f32x2){(float)rstd, (float)(-mean * rstd)};
        }
    } else
        load_aff(s_frame);
    __syncthreads();
    {
        pp_steps<0, PRND>([&](auto E) { W6_LOAD_PIECE(0, decltype(E)::value) });
        ss = aff[s_tab * 320 + s_ch * KC + chl];
        expand_mask(vmask);
        pp_steps<0, PRND>([&](auto E) { W6_NORM_PIECE(0, decltype(E)::value, 0) W6_WRITE_PIECE(0, decltype(E)::value, il) });
        advance();
        __syncthreads(); // a new frame's table (if the second chunk is already there)
        pp_steps<0, PRND>([&](auto E) { W6_LOAD_PIECE(0, decltype(E)::value) });
        ss = aff[s_tab * 320 + s_ch * KC + chl];
        expand_mask(vmask);
        pp_steps<0, PRND>([&](auto E) { W6_NORM_PIECE(0, decltype(E)::value, 0) W6_WRITE_PIECE(0, decltype(E)::value, il + C::LDS_IN) });
        advance();
        pp_steps<0, PRND>([&](auto E) { W6_LOAD_PIECE(0, decltype(E)::value) });
        r_tab[0] = s_tab; r_c0[0] = s_ch * KC; r_vmask[0] = vmask; r_lin[0] = s_lin;
        advance();
        pp_steps<0, PRND>([&](auto E) { W6_LOAD_PIECE(1, decltype(E)::value) });
        r_tab[1] = s_tab; r_c0[1] = s_ch * KC; r_vmask[1] = vmask; r_lin[1] = s_lin;
    }
    {
        const unsigned w0 = (unsigned)(lin0 % ncb) * block_wb;
        pp_steps<0, 4>([&](auto KB_) {
            constexpr int kb = decltype(KB_)::value;
            pp_steps<0, 9>([&](auto J) { w6_load_A<decltype(J)::value, kb>(rw, wlane, w_off(w0 + (unsigned)(kb >> 1) * chunk_wb, kb & 1, decltype(J)::value), A3[decltype(J)::value]); });
        });
    }
    pp_steps<0, PRND>([&](auto E) { w6_wait<0>(xv[0][decltype(E)::value]); w6_wait<0>(xv[1][decltype(E)::value]); });
    pp_steps<0, 9>([&](auto J) { w6_wait<0>(A3[decltype(J)::value]); });
    __syncthreads();
    read_raw(il + rbase, std::integral_constant<int, 0>{}, std::integral_constant<int, 6>{});
    row_pass();
    col_pass(B[0]);

    int buf = 0;
    const size_t out_plane = (size_t)p.Hout * p.Wout;
#if PP_W6_DIAG & 512
    unsigned long long st_top = 0, st_k0 = 0, st_k1 = 0, st_bar = 0, st_epi1 = 0, st_epi2 = 0, st_epi3 = 0, st_epi4 = 0, st_adv_t = 0, st_adv_n = 0, st_adv_s = 0, st_n = 0, st_tiles = 0, st_g[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
#endif

    for (int lin = lin0; lin < lin_end; lin += nloc) {
        const int cb = lin % ncb, tile = (lin / ncb) % ntile;
        const size_t fz = lin / (ncb * ntile);
        const int co0 = cb * C::BM;
        const int ox0 = p.rx0 + (tile % nbx) * C::PW, oy0 = p.ry0 + (tile / nbx) * C::PH;
        const unsigned wb_item = (unsigned)cb * block_wb;
        const unsigned wb_next_item = (unsigned)((lin + nloc < lin_end ? lin + nloc : lin) % ncb) * block_wb;

        // residual rows [r][y] of this wave's M-tile: requested at the top of the tile's LAST chunk (a chunk of MFMAs to land from HBM),
        // added in the epilogue (a zero-record descriptor returns zeros for a layer without a residual)
        const int opx = ox0 + 4 * tx, opy = oy0 + 4 * ty;
        const bool pix_ok = (opx < p.rx1) && (opy < p.ry1); // a 4x4 tile is inside the region or outside it (maps and regions are multiples of 4)
        constexpr unsigned W6_FAR = 0x80000000u;
        const unsigned plane_ob = (unsigned)out_plane * 4u, row_ob = (unsigned)p.Wout * 4u;
        const unsigned lb = pix_ok ? (unsigned)(co0 + WV * 16 + kq * 4) * plane_ob + (unsigned)(((size_t)opy * p.Wout + opx) * 4) : W6_FAR;
        const unsigned frame_bytes = (unsigned)((size_t)p.Cout * out_plane * 4);
        f32x4 rq[4][4];

        // one chunk = 8 channels = two k-steps; XP = chunk parity = register set of the load side and A-buffer pair (2 XP, 2 XP + 1)
        // POS: 0 = the tile's first chunk, 1 = its second, 2 = any later one
        // LAST: the tile's last chunk, a COMPILE-TIME property (the last pair of chunks is peeled off the loop below).  From the residual
        // requests at its top to their use in the epilogue the code is then one straight line: no branch, no loop exit, no merge at which
        // the register allocator could reconcile two assignments with v_mov copies -- copies of registers whose requests are still in
        // flight.  (Round 4 had `if (ch == nchunk - 1)` around them and a run-time debug branch around the epilogue; one build moved
        // the two youngest residual quads to buffer 3's registers and back at the loop exit, and a tile whose residual took longer than
        // a chunk to arrive added stale data: 1 launch in 3 .. 3000 in tools/w6_test.hip's soak mode.  tools/inflight_check.py lists
        // every copy out of a register some request of the kernel lands in.)
        auto chunk_body = [&](auto POS_, auto X_, auto LAST_, int ch) {
            constexpr int POS = decltype(POS_)::value;
            constexpr bool first_ = POS == 0;
            constexpr int X = decltype(X_)::value;
            constexpr bool LAST = decltype(LAST_)::value;
            static_assert(POS == 2 || POS == X, "a tile starts on an even chunk");
            static_assert(!LAST || X == 1, "a tile ends on an odd chunk");
#if PP_W6_DIAG & 512
            unsigned long long t0_ = 0, t1_ = 0, t2_ = 0, t3_ = 0, t4_ = 0, tg_[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
            W6_STAMP(t0_)
#endif
            const int nbuf = buf == 2 ? 0 : buf + 1, wbuf = buf == 0 ? 2 : buf - 1;
            float* ibw = il + wbuf * C::LDS_IN;
            // register set X holds chunk g+2: its (scale, shift) and in-image clamps; then the load side moves on to chunk g+4
            ss = aff[r_tab[X] * 320 + r_c0[X] + chl];
            if (r_lin[X] != mk_lin) { expand_mask(r_vmask[X]); mk_lin = r_lin[X]; }
#if PP_W6_DIAG & 512
            { unsigned long long a0_ = 0, a1_ = 0; const int lin_before_ = s_lin; W6_STAMP(a0_) advance(); W6_STAMP(a1_)
              if (s_lin != lin_before_) { st_adv_t += a1_ - a0_; st_adv_n += 1; } else st_adv_s += a1_ - a0_; }
#else
            advance();
#endif
            r_tab[X] = s_tab; r_c0[X] = s_ch * KC; r_vmask[X] = vmask; r_lin[X] = s_lin;
            // A operands requested during this chunk: those of chunk g+2 (this item's, or the next item's first two)
            const unsigned wnext = (PP_W6_DIAG & 128) ? 0u : ((ch + 2 < nchunk) ? wb_item + (unsigned)(ch + 2) * chunk_wb : wb_next_item + (unsigned)(ch + 2 - nchunk) * chunk_wb);
            const unsigned soff_x = (unsigned)(s_ch * KC) * plane_b;
            const float* raw1 = il + buf * C::LDS_IN + 4 * C::CS + rbase;  // this chunk's second channel quad
            const float* raw0n = il + nbuf * C::LDS_IN + rbase;            // the next chunk's first channel quad
            if constexpr (LAST) {
                const float* gres = p.res ? p.res + fz * p.res_fs : p.out;
                const i32x4 rres = w6_rsrc(gres, p.res ? frame_bytes : 0u);
#pragma unroll
                for (int r = 0; r < 2; ++r) // rows 0, 1 here; rows 2, 3 at the top of the epilogue, into the registers A buffer 3 gives up
#pragma unroll
                    for (int y = 0; y < 4; ++y) w6_load_x4(rq[r][y], rres, lb, (unsigned)r * plane_ob + (unsigned)y * row_ob);
            }
            __builtin_amdgcn_sched_barrier(0);
            W6_STAMP(t1_)
            pp_steps<0, 2>([&](auto KS_) {
                constexpr int ks = decltype(KS_)::value, kb = 2 * X + ks;
#if PP_W6_DIAG & 512
                if constexpr (ks == 1) { W6_STAMP(t2_) }
#endif
                const float* rawn = ks == 0 ? raw1 : raw0n;
                pp_steps<0, 9>([&](auto J) {
                    constexpr int j = decltype(J)::value;
#if PP_W6_DIAG & 512
                    if constexpr (ks == 0) { W6_STAMP(tg_[j]) }
#endif
                    // the A operands of (kb, j), requested two chunks ago: N_A younger operations (a lower bound around the epilogue,
                    // whose stores are not counted: waiting for a few more of the oldest costs nothing, they are two chunks old)
                    // Buffer 3 (VGPRs) is NOT re-requested in a tile's last chunk: its 36 registers are free for the epilogue, and its
                    // nine requests for the next tile are issued behind the epilogue instead -- before the next tile's first wait, so the
                    // counts of the tile's first chunks are the steady-state ones; only buffer 3's own wait in the tile's second chunk sees
                    // fewer younger operations.  Every wait of an odd chunk's second k-step allows for the (up to) nine missing requests.
                    constexpr int NW = (kb == 3) ? (POS == 1 ? (8 - j) + C::T + C::idx_A(1, j) : C::N_A - 9) : C::N_A;
                    if constexpr (kb == 3) w6_wait<NW>(A3[j]); else w6_wait<NW>();
                    if constexpr (kb < 3 && !(PP_W6_DIAG & (4 | 8 | 2048 | 4096))) {
                        w6_mfma4_load<j, kb, first_ && ks == 0>(B[ks][j], rw, wlane, w_off(wnext, ks, j)); // the MFMAs and the buffer's next request
                        __builtin_amdgcn_sched_barrier(0);
                    } else {
                        w6_mfma4<j, kb, first_ && ks == 0>(B[ks][j], A3[j]);
                        __builtin_amdgcn_sched_barrier(0);
                        if constexpr (kb == 3) { if constexpr (!LAST) w6_load_A<j, kb>(rw, wlane, w_off(wnext, ks, j), A3[j]); }
                        else w6_load_A<j, kb>(rw, wlane, w_off(wnext, ks, j), A3[j]);
                    }
                    // gaps: raw window rows of the next k-step behind positions 0..2, the two VALU clusters behind 4 and 6
                    if constexpr (j == 0) read_raw(rawn, std::integral_constant<int, 0>{}, std::integral_constant<int, 2>{});
                    if constexpr (j == 1) read_raw(rawn, std::integral_constant<int, 2>{}, std::integral_constant<int, 4>{});
                    if constexpr (j == 2) read_raw(rawn, std::integral_constant<int, 4>{}, std::integral_constant<int, 6>{});
                    constexpr int NL = (kb == 3) ? 9 : 0; // see above: the requests a tile's last chunk leaves out
                    if constexpr (j == 4) { row_pass(); W6_NORM_PIECE(X, ks * 2, C::n_piece(ks * 2) - NL) W6_NORM_PIECE(X, ks * 2 + 4, C::n_piece(ks * 2 + 4) - NL) }
                    if constexpr (j == 5) {
                        W6_WRITE_PIECE(X, ks * 2, ibw)
                        if constexpr (ks * 2 < PRND) w6_load_x4(xv[X][ks * 2], rin, goff[ks * 2], soff_x);
                        W6_WRITE_PIECE(X, ks * 2 + 4, ibw)
                        if constexpr (ks * 2 + 4 < PRND) w6_load_x4(xv[X][ks * 2 + 4 < PRND ? ks * 2 + 4 : 0], rin, goff[ks * 2 + 4 < PRND ? ks * 2 + 4 : 0], soff_x);
                    }
                    if constexpr (j == 6) { col_pass(B[ks ^ 1]); W6_NORM_PIECE(X, ks * 2 + 1, C::n_piece(ks * 2 + 1) - NL) W6_NORM_PIECE(X, ks * 2 + 5, C::n_piece(ks * 2 + 5) - NL) }
                    if constexpr (j == 7) {
                        W6_WRITE_PIECE(X, ks * 2 + 1, ibw)
                        if constexpr (ks * 2 + 1 < PRND) w6_load_x4(xv[X][ks * 2 + 1 < PRND ? ks * 2 + 1 : 0], rin, goff[ks * 2 + 1 < PRND ? ks * 2 + 1 : 0], soff_x);
                        W6_WRITE_PIECE(X, ks * 2 + 5, ibw)
                        if constexpr (ks * 2 + 5 < PRND) w6_load_x4(xv[X][ks * 2 + 5 < PRND ? ks * 2 + 5 : 0], rin, goff[ks * 2 + 5 < PRND ? ks * 2 + 5 : 0], soff_x);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                });
            });
            W6_STAMP(t3_)
            __syncthreads();
            W6_STAMP(t4_)
#if PP_W6_DIAG & 512
            st_top += t1_ - t0_; st_k0 += t2_ - t1_; st_k1 += t3_ - t2_; st_bar += t4_ - t3_; st_n += 1;
#pragma unroll
            for (int g_ = 0; g_ < 8; ++g_) st_g[g_] += tg_[g_ + 1] - tg_[g_];
            st_g[8] += t2_ - tg_[8];
#endif
            buf = nbuf;
        };
        using X0 = std::integral_constant<int, 0>;
        using X1 = std::integral_constant<int, 1>;
        using P0 = std::integral_constant<int, 0>;
        using P1 = std::integral_constant<int, 1>;
        using P2 = std::integral_constant<int, 2>;
        using LN = std::false_type;
        using LY = std::true_type;
        // nchunk >= 4 (Cin >= 32, wino6_menu): first pair, the pairs in between, last pair
        chunk_body(P0{}, X0{}, LN{}, 0);
        chunk_body(P1{}, X1{}, LN{}, 1);
#pragma unroll 1
        for (int ch = 2; ch < nchunk - 2; ch += 2) {
            chunk_body(P2{}, X0{}, LN{}, ch);
            chunk_body(P2{}, X1{}, LN{}, ch + 1);
        }
        chunk_body(P2{}, X0{}, LN{}, nchunk - 2);
        chunk_body(P2{}, X1{}, LY{}, nchunk - 1);

        // ---------------- epilogue ----------------
#if PP_W6_DIAG & 512
        unsigned long long e0_ = 0, e1_ = 0, e2_ = 0, e3_ = 0;
        W6_STAMP(e0_)
#endif
        if constexpr (!(PP_W6_DIAG & 16384)) {
            // an 8-pass MFMA's D needs 12 wait states before anything but the next accumulating MFMA touches it (hipcc pads nothing
            // behind an asm statement)
            asm volatile("s_nop 11" ::: W6_AGPRS);
            float* __restrict__ gout = p.out + fz * p.out_fs;
            const i32x4 rout = w6_rsrc(gout, frame_bytes);
            {
                const float* gres = p.res ? p.res + fz * p.res_fs : p.out;
                const i32x4 rres = w6_rsrc(gres, p.res ? frame_bytes : 0u);
#pragma unroll
                for (int r = 2; r < 4; ++r)
#pragma unroll
                    for (int y = 0; y < 4; ++y) w6_load_x4(rq[r][y], rres, lb, (unsigned)r * plane_ob + (unsigned)y * row_ob);
            }

            // column half of the output transform on this wave's positions, per M-tile and accumulator row pair:
            //   full row (m0..m5)  ->  T = (m0 + s12 + s34, d12 + 2 d34, s12 + 4 s34, d12 + 8 d34 + m5)
            //   left half (m0, m1, m2) -> (m0 + s12, d12, s12, d12)        right half (m3, m4, m5) -> (s34, 2 d34, 4 s34, 8 d34 + m5)
            f32x2 ownF[4][2], ownH[4][2]; // this wave's own M-tile: [x][row pair]
            pp_steps<0, 4>([&](auto MT_) {
                constexpr int mt = decltype(MT_)::value;
                constexpr int src = WV - (WV > mt ? 1 : 0); // slot of this wave among the three sources of M-tile mt
                pp_steps<0, 2>([&](auto RP_) {
                    constexpr int rp = decltype(RP_)::value;
                    f32x2 mm[9];
                    w6_acc_read<mt, rp>(mm);
                    const f32x2 s12 = mm[1] + mm[2], d12 = mm[1] - mm[2], s34 = mm[3] + mm[4], d34 = mm[3] - mm[4];
                    f32x2 tf[4], th[4];
                    tf[0] = mm[0] + s12 + s34;
                    tf[1] = pkfma(d34, 2.f, d12);
                    tf[2] = pkfma(s34, 4.f, s12);
                    tf[3] = pkfma(d34, 8.f, d12) + mm[5];
                    if constexpr ((WV & 1) == 0) {
                        const f32x2 hs = mm[7] + mm[8], hd = mm[7] - mm[8];
                        th[0] = mm[6] + hs; th[1] = hd; th[2] = hs; th[3] = hd;
                    } else {
                        const f32x2 hs = mm[6] + mm[7], hd = mm[6] - mm[7];
                        th[0] = hs; th[1] = hd + hd; th[2] = hs * (f32x2){4.f, 4.f}; th[3] = pkfma(hd, 8.f, mm[8]);
                    }
                    if constexpr (mt == WV) {
#pragma unroll
                        for (int x = 0; x < 4; ++x) { ownF[x][rp] = tf[x]; ownH[x][rp] = th[x]; }
                    } else if constexpr (!(PP_W6_DIAG & 32)) {
                        float* wf = xb + ((((mt * 3 + src) * 2 + 0) * 4 + 2 * rp) * 64 + lane) * 4;
                        float* wh = xb + ((((mt * 3 + src) * 2 + 1) * 4 + 2 * rp) * 64 + lane) * 4;
                        *reinterpret_cast<f32x4*>(wf) = (f32x4){tf[0][0], tf[1][0], tf[2][0], tf[3][0]};
                        *reinterpret_cast<f32x4*>(wf + 256) = (f32x4){tf[0][1], tf[1][1], tf[2][1], tf[3][1]};
                        *reinterpret_cast<f32x4*>(wh) = (f32x4){th[0][0], th[1][0], th[2][0], th[3][0]};
                        *reinterpret_cast<f32x4*>(wh + 256) = (f32x4){th[0][1], th[1][1], th[2][1], th[3][1]};
                    }
                });
            });
            // the residual rows were requested a whole chunk ago (T operations younger than the last of them)
            W6_STAMP(e1_)
            __syncthreads();
            W6_STAMP(e2_)
            // row half for M-tile WV: T[i] (float4 over x) from the four waves, Y[y] = sum_i A^T[y][i] T[i]
            float ssum[4], ssq[4];
            pp_steps<0, 4>([&](auto R_) {
                constexpr int r = decltype(R_)::value;
                f32x4 T[6], TH[2][2]; // TH[row 1 | row 4][left | right]
                auto own4 = [&](const f32x2 (&o)[4][2]) { return (f32x4){o[0][r >> 1][r & 1], o[1][r >> 1][r & 1], o[2][r >> 1][r & 1], o[3][r >> 1][r & 1]}; };
                pp_steps<0, 4>([&](auto W_) {
                    constexpr int w = decltype(W_)::value;
                    constexpr int irow = w == 0 ? 0 : w == 1 ? 2 : w == 2 ? 3 : 5;
                    if constexpr (w == WV) { T[irow] = own4(ownF); TH[w >> 1][w & 1] = own4(ownH); }
                    else {
                        constexpr int src = w - (w > WV ? 1 : 0);
                        if constexpr (!(PP_W6_DIAG & 32)) {
                            T[irow] = *reinterpret_cast<const f32x4*>(xb + ((((WV * 3 + src) * 2 + 0) * 4 + r) * 64 + lane) * 4);
                            TH[w >> 1][w & 1] = *reinterpret_cast<const f32x4*>(xb + ((((WV * 3 + src) * 2 + 1) * 4 + r) * 64 + lane) * 4);
                        } else { T[irow] = own4(ownF); TH[w >> 1][w & 1] = own4(ownH); }
                    }
                });
                T[1] = TH[0][0] + TH[0][1];
                T[4] = TH[1][0] + TH[1][1];
                const f32x4 s12 = T[1] + T[2], d12 = T[1] - T[2], s34 = T[3] + T[4], d34 = T[3] - T[4];
                f32x4 Y[4];
                Y[0] = T[0] + s12 + s34;
                Y[1] = __builtin_elementwise_fma(d34, (f32x4){2.f, 2.f, 2.f, 2.f}, d12);
                Y[2] = __builtin_elementwise_fma(s34, (f32x4){4.f, 4.f, 4.f, 4.f}, s12);
                Y[3] = __builtin_elementwise_fma(d34, (f32x4){8.f, 8.f, 8.f, 8.f}, d12) + T[5];
                f32x4 sv = {0.f, 0.f, 0.f, 0.f}, qv = {0.f, 0.f, 0.f, 0.f};
                // residual rows of r.  vmcnt orders loads among loads and stores among stores, NOT one against the other: a younger store that
                // completes first lowers the counter, so a count that allows for pending stores proves nothing about an older load (the first
                // version waited for rows 2, 3 with "the 4 r stores issued so far" and, once in ~1000 frames, added a residual row that had
                // not landed: tools/perm_probe.py).  Hence NO store is issued before the last residual wait -- the finished rows stay in the
                // residual's registers and go out together below -- and the counts are the younger LOADS alone: rows 0, 1 -- the last chunk's
                // T - 9 requests and the 8 requests of rows 2, 3 (a lower bound for all but the youngest quad); rows 2, 3 -- the quads behind
                pp_steps<0, 4>([&](auto Y_) {
                    constexpr int y = decltype(Y_)::value;
                    w6_wait<(r < 2 ? C::T - 9 + 8 : 7 - (4 * (r - 2) + y))>(rq[r][y]);
                });
#pragma unroll
                for (int y = 0; y < 4; ++y) {
                    const f32x4 v = Y[y] + rq[r][y];
                    rq[r][y] = v;
                    sv += v;
                    qv = __builtin_elementwise_fma(v, v, qv);
                }
                ssum[r] = pix_ok ? (sv[0] + sv[1]) + (sv[2] + sv[3]) : 0.f;
                ssq[r] = pix_ok ? (qv[0] + qv[1]) + (qv[2] + qv[3]) : 0.f;
            });
            if constexpr (!(PP_W6_DIAG & 1024)) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int y = 0; y < 4; ++y)
                        asm volatile(W6_SGPR_PAD "buffer_store_dwordx4 %0, %1, %2, %3 offen\n\ts_nop 1" :: "v"(rq[r][y]), "v"(lb), "s"(rout), "s"((unsigned)r * plane_ob + (unsigned)y * row_ob) : "memory");
            }
#if PP_W6_DIAG & 512
            { unsigned long long e2b_ = 0; W6_STAMP(e2b_) st_epi4 += e2b_ - e2_; }
#endif
            if (p.stat_acc) {
                // ONE atomic instruction per wave and tile: after the row reductions every lane of a 16-lane row holds the row's four (sum, sum of
                // squares) pairs; lane m < 8 of each row adds component m & 1 of accumulator row m >> 1, so the 32 values of the wave's 16
                // channels go out as 32 lanes of one global_atomic_add_f64 (eight instructions of 4 lanes each cost 0.2 ms of a 2 ms launch:
                // a CU retires about one atomic wave-instruction per 50 ns whatever its lane count)
                float sel[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float s = row16_sum(ssum[r]), q = row16_sum(ssq[r]);
                    sel[r] = (m & 1) ? q : s;
                }
                const int rr = (m >> 1) & 3;
                const float v = rr == 0 ? sel[0] : rr == 1 ? sel[1] : rr == 2 ? sel[2] : sel[3];
                if (m < 8) {
                    double* dst = p.stat_acc + fz * p.stat_fs + ((size_t)(blockIdx.x % NREP) * p.stat_C + co0 + WV * 16 + kq * 4 + rr) * 2 + (m & 1);
                    const double dv = (double)v;
                    asm volatile("global_atomic_add_f64 %0, %1, off\n\ts_nop 1" :: "v"(dst), "v"(dv) : "memory");
                }
            }
        } else {
            pp_steps<0, 2>([&](auto R_) { pp_steps<0, 4>([&](auto Y_) { w6_wait<C::T - 9>(rq[decltype(R_)::value][decltype(Y_)::value]); }); });
        }
        // buffer 3 for the next tile's second chunk (see the chunk body)
        pp_steps<0, 9>([&](auto J) { w6_load_A<decltype(J)::value, 3>(rw, wlane, w_off(wb_next_item + chunk_wb, 1, decltype(J)::value), A3[decltype(J)::value]); });
#if PP_W6_DIAG & 512
        W6_STAMP(e3_)
        st_epi1 += e1_ - e0_; st_epi2 += e2_ - e1_; st_epi3 += e3_ - e2_; st_tiles += 1;
#endif
    }
#if PP_W6_DIAG & 512
    if (lane == 0 && p.dbg_buf) {
        unsigned long long* q = p.dbg_buf + WV * 24;
        atomicAdd(q + 0, st_top); atomicAdd(q + 1, st_k0); atomicAdd(q + 2, st_k1); atomicAdd(q + 3, st_bar); atomicAdd(q + 4, st_epi1); atomicAdd(q + 5, st_epi2);
        atomicAdd(q + 6, st_epi3); atomicAdd(q + 7, st_n); atomicAdd(q + 8, st_tiles);
        for (int g_ = 0; g_ < 9; ++g_) atomicAdd(q + 9 + g_, st_g[g_]);
        atomicAdd(q + 18, st_epi4); atomicAdd(q + 19, st_adv_t); atomicAdd(q + 20, st_adv_n); atomicAdd(q + 21, st_adv_s);
    }
#endif
    w6_wait<0>(); // nothing of this wave's requests may land in registers a successor workgroup owns
#undef W6_LOAD_PIECE
#undef W6_NORM_PIECE
#undef W6_WRITE_PIECE
}

template <int TWT, int ROOF>
__global__ void __launch_bounds__(256, 1) wino6_mfma(const ConvP p)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if (wv == 0) wino6_body<TWT, 0, ROOF>(p, smem);
    else if (wv == 1) wino6_body<TWT, 1, ROOF>(p, smem);
    else if (wv == 2) wino6_body<TWT, 2, ROOF>(p, smem);
    else wino6_body<TWT, 3, ROOF>(p, smem);
}

template <int TWT>
Variant make_wino6(bool roofline_layer)
{
    using C = Wino6Cfg<TWT>;
    Variant v;
    v.kern = roofline_layer ? wino6_mfma<TWT, 1> : wino6_mfma<TWT, 0>;
    v.bm = C::BM; v.bmp = C::BM; v.pw = C::PW; v.ph = C::PH; v.kc = C::KC; v.threads = C::THREADS;
    v.waves = 4; v.pairs = 4 * 9;
    v.lds = (size_t)C::LDS_FLOATS * sizeof(float);
    v.wino = 6;
    snprintf(v.name, sizeof(v.name), "wino6 tw%d", TWT);
    return v;
}

} // namespace

// In the tuner's menu by default (PP_WINO6=0 removes it).  Measured on MI355X (profiles/r04_wino6_ablation.txt, r04 bench): slower than wino4_mfma on
// the 64-channel 400 x 400 layers (952 against 852 us per 16-frame launch), faster on the 128-channel 200 x 200 and 256-channel 100 x 100 layers
// (710 / 757 us per 16-frame launch, 72 / 70 against 82 / 83 us at one frame per launch), where a tile carries 16 / 32 chunks per epilogue -- the
// tuner decides per layer.
void wino6_menu(std::vector<Variant>& menu, bool roofline_layer)
{
    const char* e = getenv("PP_WINO6"); // read per call (commit time only)
    if (!(e && e[0] == '0')) menu.push_back(make_wino6<4>(roofline_layer)); // 16 x 16 px
}
// strip tilings of the region launches (launch_conv): a map that is no multiple of 16 x 16 is covered by whole main tiles plus thin tiles
Variant wino6_strip_v() { return make_wino6<1>(false); }  // 4 px wide, 64 px tall
Variant wino6_strip_h() { return make_wino6<16>(false); } // 64 px wide, 4 px tall

// Transformed weights U = G g G^T (fp64 on the host, rounded once) in the order the waves fetch them:
//   [cout block][k-step][wave][local position][lane = (cin quad lane kq) * 16 + m][M-tile]  =  U[block*64 + mt*16 + m][4 s + kq][i][jj]
void wino6_pack(const float* w /*[rows][cin][3][3]*/, int rows, int cin, std::vector<float>& out)
{
    static const double G[6][3] = {{1.0 / 4, 0, 0}, {-1.0 / 6, -1.0 / 6, -1.0 / 6}, {-1.0 / 6, 1.0 / 6, -1.0 / 6},
                                   {1.0 / 24, 1.0 / 12, 1.0 / 6}, {1.0 / 24, -1.0 / 12, 1.0 / 6}, {0, 0, 1}};
    const int nblk = rows / 64, nks = cin / 4;
    out.assign((size_t)rows * cin * 36, 0.f);
    std::vector<float> u((size_t)36);
    for (int row = 0; row < rows; ++row)
        for (int c = 0; c < cin; ++c) {
            const float* g = w + ((size_t)row * cin + c) * 9;
            double t[6][3];
            for (int a = 0; a < 6; ++a)
                for (int j = 0; j < 3; ++j) t[a][j] = G[a][0] * g[0 * 3 + j] + G[a][1] * g[1 * 3 + j] + G[a][2] * g[2 * 3 + j];
            for (int a = 0; a < 6; ++a)
                for (int b = 0; b < 6; ++b) u[a * 6 + b] = (float)(t[a][0] * G[b][0] + t[a][1] * G[b][1] + t[a][2] * G[b][2]);
            const int blk = row / 64, mt = (row % 64) / 16, mm = row % 16, s = c / 4, kq = c % 4;
            for (int wv = 0; wv < 4; ++wv)
                for (int j = 0; j < 9; ++j) {
                    const size_t o = ((((((size_t)blk * nks + s) * 4 + wv) * 9 + j) * 64) + kq * 16 + mm) * 4 + mt;
                    out[o] = u[w6_pos_i(wv, j) * 6 + w6_pos_j(wv, j)];
                }
        }
    (void)nblk;
}

} // namespace ppc
