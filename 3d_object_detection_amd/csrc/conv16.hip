// 3x3 convolutions (stride 1 and 2) with 16-bit MFMA operands -- the reduced-precision deploy modes of SURVEY 8(f).4.
// The reference deploys the whole RPN as a TensorRT FP16 engine (framework/trt_utils.py:30,
// networks/pointpillars8_trt.py:208-223); here: fp16, bf16 and split-bf16 ("bf16x3") operands on
// v_mfma_f32_32x32x16_{f16,bf16}, fp32 accumulation, activations fp32 NCHW in HBM exactly as on the fp32 path (so every
// intermediate stays comparable and the modes can be mixed per layer).
//
//   D[cout, pixel] = sum_{tap, cin} W[cout, tap, cin] * relu(norm(X[cin, pixel + tap]))
//
// * Direct convolution, no Winograd: at 16x the fp32-MFMA rate the matrix pipe is no longer what bounds these layers -- the
//   64-channel layers at 400 x 400 are HBM-bound (82 MB in + out per frame against 11.8 GFLOP), the 256-channel ones at
//   100 x 100 MFMA-bound, the 128-channel ones in between.
// * Persistent workgroups over (frame, tile, cout block) items with XCD-contiguous ranges, in three residency shapes the
//   tuner chooses between: o1 = one 4-wave workgroup per CU (one wave per SIMD, the whole register file, 64 x 160-pixel wave
//   tiles, double-buffered LDS); o2 = two 4-wave workgroups per CU, or ONE 8-wave workgroup (w4x2 / w2x4: one weight image
//   and one patch per 320 / 640 pixels) -- two waves per SIMD, <= 256 registers, 32 x 160-pixel wave tiles, single-buffered
//   LDS.  A workgroup tile is PW x PH output pixels = 32 * WN * NT pixels taken in row-major
//   order: N-tile j of the tile = pixels 32 j .. 32 j + 31 of that order, so ANY tile width that divides the map works
//   (400 = 5 * 80, 200 = 5 * 40, 100 = 5 * 20) and a wave's stores are runs of up to PW contiguous pixels of one channel.
// * Per 16 input channels ("step"): the [IH][COLS] halo patch is fetched as aligned dwordx4 (4 pixels) of 8 channels per
//   thread, normalised (x * scale + shift, ReLU; scale / shift of the producer's InstanceNorm or folded BatchNorm arrive
//   as scalar loads: a wave's threads all stage the same channel octet), zero-padded, rounded to the operand type and
//   written to LDS as [k-half][row][col][8 channels] -- one 16-byte unit per pixel and k-half, which IS the B operand of
//   one lane.  Row stride == PW (mod 16 units): a wave's 32 lanes then read 32 consecutive units modulo 16 -> conflict-free
//   ds_read_b128 for every tap shift.  Stride 2: even / odd columns de-interleaved (the tile walk stays unit-stride).
//   The weights of the step, pre-packed on the host in the LDS image [tap][k-half][row][8], are a linear copy.
// * Double-buffered LDS, one barrier per step, ONE staging register set written after the barrier and re-requested at once:
//   at the top of step s the data of step s+1 (requested a whole step earlier) is written to the other buffer and the
//   loads of step s+2 are requested right behind it, then the 9 * MT * NT MFMAs of step s run.
// * bf16x3: x = hi + lo (hi = bf16(x), lo = bf16(x - hi)), x * w ~ hi*hi + hi*lo + lo*hi as THREE steps per 16 channels
//   with the same LDS footprint, 3x the MFMAs of bf16 -- fp32-equivalent for this network (DESIGN.md tolerance table).
//   Double-buffered path: patch and weight buffers are indexed separately, so a step stages only the image that changes
//   (x_hi + w_hi, then w_lo, then x_lo: four images per chunk; the fp32 patch comes from L2 the second time); the
//   single-buffered path re-stages both images every step.
// * Epilogue: residual add, fp32 NCHW stores through a buffer descriptor (lane offset in a VGPR, the channel's plane offset
//   in an SGPR, out-of-map lanes parked beyond the descriptor's range), per-channel sum / sum of squares for the consumer's
//   InstanceNorm (fp32 over the wave's <= NT*32 pixels, then fp64 atomics into the 8 replicated accumulators).
#include <cstdio>
#include <cstring>
#include "conv_common.h"

namespace ppc {
namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

enum { P16_BF16X3 = 1, P16_BF16 = 2, P16_FP16 = 3 };

// diagnostic build (make stamp16; tools/c16_stamp.py): s_memtime sums of the phases of an item, lane 0 of wave 0 of every
// workgroup, into ConvP::dbg_buf[0..7] = prologue | loads issued | tap loop | barrier after the taps | commit + rest of staging |
// second barrier | epilogue | items
#ifndef C16_STAMP
#define C16_STAMP 0
#endif
#if C16_STAMP
#define C16_T(x) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); st_[x] += t_ - tl_; tl_ = t_; }
#else
#define C16_T(x)
#endif

template <int PREC>
__device__ __forceinline__ unsigned pack2(float a, float b)
{
    if constexpr (PREC == P16_FP16) return __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){a, b}, f16x2));
    else return __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){a, b}, bf16x2));
}
// the part of (a, b) a bf16x3 step stages: part 0 = hi, part 1 = lo = bf16(x - hi)
__device__ __forceinline__ unsigned pack2_lo(float a, float b)
{
    const unsigned h = pack2<P16_BF16>(a, b);
    return pack2<P16_BF16>(a - __uint_as_float(h << 16), b - __uint_as_float(h & 0xFFFF0000u));
}

template <int PREC>
__device__ __forceinline__ f32x16 mfma16(const u32x4 a, const u32x4 b, const f32x16 c)
{
    if constexpr (PREC == P16_FP16)
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

constexpr int ceil_div(int a, int b) { return (a + b - 1) / b; }

template <int STRIDE, int WM, int WN, int MT, int NT, int PW, int PH, int OCC = 1>
struct C16 {
    static constexpr bool DB = (OCC == 1); // one workgroup per CU: double-buffered LDS; two per CU: single buffer, two barriers per step
    static constexpr int S = STRIDE;
    static constexpr int BM = WM * MT * 32;
    static constexpr int NPIX = PW * PH;
    static_assert(NPIX == 32 * WN * NT, "tile = whole N-tiles of 32 pixels");
    static constexpr int NW = WM * WN;   // waves per workgroup: 4 (one or two workgroups per CU) or 8 (one workgroup, two waves per SIMD)
    static constexpr int T = 64 * NW;    // threads
    static constexpr int TH = T / 2;     // threads staging one channel octet
    static_assert(NW == 4 || NW == 8, "four or eight waves");
    static_assert(NW == 4 || OCC == 2, "eight waves = two per SIMD: the register budget of OCC 2, single-buffered LDS");
    static constexpr int WGS_PER_CU = OCC * 4 / NW;
    static_assert((PW * S) % 4 == 0, "tile origin on a pixel quad of the input");
    static constexpr int IH = (PH - 1) * S + 3;
    static constexpr int NQ = ceil_div((PW - 1) * S + 6, 4); // aligned pixel quads per patch row, from x = ox0*S - 4
    static constexpr int COLS = NQ * 4;
    static constexpr int HALF = COLS / 2; // stride 2: even columns [0, HALF), odd columns [HALF, COLS)
    // row stride in 16-byte units: lane n of an N-tile reads unit (row(n) * S) * RS + col(n); consecutive pixels of the
    // row-major tile order must land on consecutive units modulo 16
    static constexpr int rs()
    {
        int v = COLS;
        while ((S * v) % 16 != PW % 16) ++v;
        return v;
    }
    static constexpr int RS = rs();
    static_assert(S == 1 || PW % 2 == 0, "stride 2: S*RS == PW (mod 16) needs an even tile width");
    static constexpr int X_UNITS = 2 * IH * RS;  // [k-half][row][col]
    static constexpr int W_UNITS = 9 * 2 * BM;   // [tap][k-half][row]
    static constexpr int POS = IH * NQ;          // (row, quad) positions per channel octet
    static constexpr int XR = ceil_div(POS, TH); // staging rounds: the first half of the waves stages octet 0, the second half octet 1
    static constexpr int WR = ceil_div(W_UNITS, T);
    static constexpr int RED_FLOATS = WN * BM * 2;
    static constexpr int SCR_FLOATS = NW * 32 * 36; // wave-private transpose tiles of the epilogue
    // OCC 2: the transpose tiles alias the (idle) operand buffers; a barrier closes every item
    static_assert(DB || (size_t)(X_UNITS + W_UNITS) * 16 >= (size_t)SCR_FLOATS * 4, "transpose tiles must fit the operand buffers");
    // single-buffered patch, but TWO weight images where the CU's LDS allows it: the weights of step s+1 are requested before the
    // MFMA loop of step s and written behind it, ahead of the barrier (the image a workgroup re-reads per step is the larger half of
    // what it stages, and fetched behind the barrier its latency and its trip through the memory pipe were exposed)
    static constexpr bool WDB = !DB && ((size_t)(X_UNITS + 2 * W_UNITS) * 16 + RED_FLOATS * 4) * WGS_PER_CU <= 160 * 1024;
    static constexpr int NWBUF = (DB || WDB) ? 2 : 1;
    static constexpr size_t LDS_BYTES = DB ? (size_t)(2 * (X_UNITS + W_UNITS)) * 16 + (RED_FLOATS + SCR_FLOATS) * 4
                                           : (size_t)(X_UNITS + NWBUF * W_UNITS) * 16 + RED_FLOATS * 4;
    static_assert(PW % 4 == 0, "the epilogue stores pixel quads");
};

// sparse first layer (stride 2 from the BEV grid): positions are single pixels, channels come from the PFN rows
// SPARSE (stride 2 only): the first conv of the fused path -- the input is the pillar-index map of the BEV grid plus the PFN
// rows (ConvP::pmap / feat) instead of a dense canvas; a tile whose halo patch holds no pillar skips its MFMA loop.  A twin
// instantiation, so that the dense stride-2 layers do not carry its registers.
// IO16 (pp_set_precision 4, "fp16s"): bit 0 = the input tensor is fp16, bit 1 = the output tensor (and the residual, which has the
// output's layout) is fp16.  The arithmetic is the fp16-operand path either way: fp16 input is widened as it arrives and goes
// through the same fp32 normalise / ReLU / round; outputs are rounded to fp16 behind the statistics.
template <int STRIDE, int PREC, int WM, int WN, int MT, int NT, int PW, int PH, int OCC, bool SPARSE = false, int IO16 = 0>
__global__ void __launch_bounds__(64 * WM * WN, OCC) conv16(const ConvP p)
{
    constexpr bool IN16 = (IO16 & 1) != 0, OUT16 = (IO16 & 2) != 0;
    static_assert(!(SPARSE && IN16), "the sparse first conv reads the fp32 PFN rows");
    constexpr unsigned EBI = IN16 ? 2u : 4u, EBO = OUT16 ? 2u : 4u; // bytes per element of the input / output tensor
    using C = C16<STRIDE, WM, WN, MT, NT, PW, PH, OCC>;
    constexpr int NBUF = C::DB ? 2 : 1;
    constexpr int S = STRIDE;
    constexpr int PASSES = (PREC == P16_BF16X3) ? 3 : 1;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    u32x4* xl = reinterpret_cast<u32x4*>(smem_raw);               // [NBUF][X_UNITS]
    u32x4* wl = xl + NBUF * C::X_UNITS;                            // [NWBUF][W_UNITS]
    float* red = reinterpret_cast<float*>(wl + C::NWBUF * C::W_UNITS); // [WN][BM][2]
    float* scr_all = C::DB ? red + C::RED_FLOATS : reinterpret_cast<float*>(smem_raw); // [4 waves][32][36] epilogue transpose tiles

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int n32 = lane & 31, kh = lane >> 5;
    const int oct = wave / (C::NW / 2);        // channel octet this wave stages (uniform)
    const int st = tid - oct * C::TH;          // staging thread index inside the octet's threads

    const int tiles_x = (p.Wout + PW - 1) / PW, tiles_y = (p.Hout + PH - 1) / PH;
    const int ncb = p.Cout / C::BM;
    const int per_frame = tiles_x * tiles_y * ncb;
    const int total = per_frame * p.nb;
    // XCD-contiguous item ranges: blocks b, b + 8, ... share an XCD (round-robin placement) and walk ONE eighth of the
    // (frame, tile row, tile column, cout block) order, so neighbouring tiles (shared halo rows) and the cout blocks of a
    // tile (same patch) meet in one L2.  Placement is a speed matter only.
    const int nslots = gridDim.x >> 3, xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int per_xcd = (total + 7) >> 3;
    // diagnostics (PP_CONV_DBG, timing only -- results are wrong): 1 stage only step 0 of an item, 4 skip the epilogue,
    // 16 skip the step loop, 128 skip the commit (loads stay).  (No hook inside the tap loop: a branch there moves the accumulators out of the AGPRs.)
    const int nsteps = (p.dbg & 16) ? 0 : (p.Cin / 16) * PASSES;
    constexpr bool sparse = SPARSE;
    static_assert(!SPARSE || S == 2, "the sparse BEV input feeds the stride-2 first conv");
    const size_t in_plane = (size_t)p.Hin * p.Win, out_plane = (size_t)p.Hout * p.Wout;

    // lane constants of the MFMA side: LDS unit of this lane's pixel for each of its N-tiles (tap (0,0), k-half kh)
    int boff[NT], pix_off[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int pp = (wn * NT + j) * 32 + n32;
        const int py = pp / PW, px = pp - py * PW;
        // input column of output px for tap kx: px*S + kx - 1 = patch column px*S + kx + 3 (the patch starts at ox0*S - 4).
        // stride 1: unit px + 3 (+ kx per tap); stride 2: plane (kx+3)&1, unit px + (kx+3)/2 -- the tap part is added per tap
        boff[j] = kh * (C::IH * C::RS) + (py * S) * C::RS + (S == 1 ? px + 3 : px);
        pix_off[j] = py * 65536 + px; // decoded in the epilogue
    }
    const int aoff = kh * C::BM + wm * MT * 32 + n32;

#if C16_STAMP
    unsigned long long st_[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long tl_ = __builtin_amdgcn_s_memtime();
#endif
    for (int it = slot; it < per_xcd; it += nslots) {
        const int item = xcd * per_xcd + it;
        if (item >= total) break;
        const int fr = item / per_frame;
        int r_ = item - fr * per_frame;
        const int cb = r_ % ncb;
        r_ /= ncb;
        const int tx = r_ % tiles_x, ty = r_ / tiles_x;
        const int ox0 = tx * PW, oy0 = ty * PH;
        const int co0 = cb * C::BM;
        const int iy0 = oy0 * S - 1, qx0 = ox0 * S - 4;
        const float* __restrict__ gin = IN16 ? reinterpret_cast<const float*>(reinterpret_cast<const _Float16*>(p.in) + (size_t)fr * p.in_fs) : p.in + (size_t)fr * p.in_fs;
        __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(gin), 0, 0x7FFFFFFF, 0x00020000);
        const unsigned plane_b = (unsigned)(in_plane * EBI);
        const int32_t* __restrict__ gmap = sparse ? p.pmap + (size_t)fr * p.pmap_fs : nullptr;
        const float* __restrict__ gfeat = sparse ? p.feat + (size_t)fr * p.feat_fs : nullptr;
        const u32x4* __restrict__ wsrc = reinterpret_cast<const u32x4*>(p.w) + (size_t)cb * (p.Cin / 16) * ((PREC == P16_BF16X3) ? 2 : 1) * C::W_UNITS;

        // ---- staging map of this item: (row, quad) positions of this thread's rounds ----
        int goff[C::XR];      // element offset inside a channel plane (dense) / unused (sparse)
        int loff[C::XR];      // LDS unit of pixel 0 of the quad inside the k-half plane; -1 = no position
        unsigned okm = 0u;    // bit r: the quad lies inside the map (else zero padding)
#pragma unroll
        for (int r = 0; r < C::XR; ++r) {
            const int pos = st + r * C::TH;
            const int row = pos / C::NQ, q = pos - row * C::NQ;
            const int gy = iy0 + row, gx = qx0 + 4 * q;
            const bool have = pos < C::POS;
            const bool ok = have && gy >= 0 && gy < p.Hin && gx >= 0 && gx < p.Win; // Win % 4 == 0: a quad is inside or outside as a whole
            goff[r] = ok ? gy * p.Win + gx : 0;
            okm |= (ok ? 1u : 0u) << r;
            loff[r] = have ? row * C::RS + (S == 1 ? 4 * q : 2 * q) : -1; // stride 2: pixels (0,2) -> even plane 2q, 2q+1; (1,3) -> odd plane
        }
        // sparse input: pillar ids of the quads' 4 pixels
        int pid[C::XR][4];
        int any = 1;
        if (sparse) {
            any = 0;
#pragma unroll
            for (int r = 0; r < C::XR; ++r) {
#pragma unroll
                for (int e = 0; e < 4; ++e) pid[r][e] = -1;
                if ((okm >> r) & 1u) {
                    const int4 m4 = *reinterpret_cast<const int4*>(gmap + goff[r]);
                    pid[r][0] = m4.x; pid[r][1] = m4.y; pid[r][2] = m4.z; pid[r][3] = m4.w;
                    any |= (m4.x >= 0) | (m4.y >= 0) | (m4.z >= 0) | (m4.w >= 0);
                }
            }
            any = __syncthreads_or(any);
        }

        f32x16 acc[MT][NT];
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

        // staging registers: one workgroup per CU keeps a whole step in flight under the MFMAs (XR rounds + the weight image);
        // two per CU keep ONE round (the rest is fetched and written behind the step's barrier, under the partner's MFMAs)
        constexpr int XRL = C::DB ? C::XR : 1;
        f32x4 xv[XRL][8];
        u32x4 wv[C::WR];
        float sc8[8], sh8[8];

        // step -> (channel chunk, bf16x3 pass).  pass 0: x_hi * w_hi, 1: x_hi * w_lo, 2: x_lo * w_hi
        auto issue_x = [&](int step, auto R, auto SLOT) __attribute__((always_inline)) { // round R of the step's patch -> xv[SLOT]
            constexpr int r = decltype(R)::value, sl = decltype(SLOT)::value;
            const int c0 = (step / PASSES) * 16 + oct * 8;
            if (sparse) {
#pragma unroll
                for (int e = 0; e < 4; ++e) { // xv[.][2e], xv[.][2e+1] = channels 0-3, 4-7 of pixel e
                    const float* row = gfeat + (size_t)(pid[r][e] >= 0 ? pid[r][e] : 0) * 64 + c0;
                    xv[sl][2 * e] = *reinterpret_cast<const f32x4*>(row);
                    xv[sl][2 * e + 1] = *reinterpret_cast<const f32x4*>(row + 4);
                }
            } else {
                // lane offset in a VGPR, the channel's plane offset in an SGPR: no address arithmetic on the VALU
#pragma unroll
                for (int c = 0; c < 8; ++c)
                    if constexpr (IN16) { // 4 pixels = 8 bytes, kept as they arrive (components 0, 1) and widened at the commit: a conversion
                                          // here would sit in front of the MFMA loop and make the wave wait for the loads it has just issued
                        const f32x2 h_ = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rin, (unsigned)goff[r] * 2u, (unsigned)(c0 + c) * plane_b, 0));
                        xv[sl][c][0] = h_[0];
                        xv[sl][c][1] = h_[1];
                    } else
                        xv[sl][c] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rin, (unsigned)goff[r] * 4u, (unsigned)(c0 + c) * plane_b, 0));
            }
        };
        auto load_aff = [&](int step) __attribute__((always_inline)) {
            if (p.pre != PRE_RAW) {
                const int c0 = (step / PASSES) * 16 + oct * 8;
                const float* scp = p.pre_scale + (size_t)fr * p.aff_fs + c0; // wave-uniform address: scalar loads
                const float* shp = p.pre_shift + (size_t)fr * p.aff_fs + c0;
#pragma unroll
                for (int c = 0; c < 8; ++c) { sc8[c] = scp[c]; sh8[c] = shp[c]; }
            }
        };
        auto issue_w = [&](int step) __attribute__((always_inline)) {
            const int ch = step / PASSES, ps = step - ch * PASSES;
            const int img = (PREC == P16_BF16X3) ? (ch * 2 + (ps == 1 ? 1 : 0)) : ch;
            const u32x4* g = wsrc + (size_t)img * C::W_UNITS;
#pragma unroll
            for (int r = 0; r < C::WR; ++r) {
                const int e = tid + r * C::T;
                wv[r] = g[e < C::W_UNITS ? e : C::W_UNITS - 1];
            }
        };
        auto commit_x = [&](int step, int buf, auto R, auto SLOT) __attribute__((always_inline)) {
            constexpr int r = decltype(R)::value, sl = decltype(SLOT)::value;
            const bool lo = (PREC == P16_BF16X3) && (step % PASSES) == 2;
            u32x4* xb = xl + buf * C::X_UNITS + oct * (C::IH * C::RS);
            if (loff[r] < 0) return;
            const bool ok = (okm >> r) & 1u;
#pragma unroll
            for (int e = 0; e < 4; ++e) { // pixel e of the quad: 8 channels -> one 16-byte unit
                float v[8];
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    float x;
                    if constexpr (IN16) x = (float)__builtin_bit_cast(f16x4, (f32x2){xv[sl][c][0], xv[sl][c][1]})[e];
                    else x = sparse ? xv[sl][2 * e + (c >> 2)][c & 3] : xv[sl][c][e];
                    if (p.pre != PRE_RAW) x = fmaxf(fmaf(x, sc8[c], sh8[c]), 0.f);
                    const bool live = sparse ? (pid[r][e] >= 0) : ok;
                    v[c] = live ? x : 0.f; // zero padding / empty cell, in the normalised domain
                }
                u32x4 u;
#pragma unroll
                for (int c = 0; c < 4; ++c) u[c] = lo ? pack2_lo(v[2 * c], v[2 * c + 1]) : pack2<PREC>(v[2 * c], v[2 * c + 1]);
                const int unit = (S == 1) ? loff[r] + e : loff[r] + (e & 1) * C::HALF + (e >> 1);
                xb[unit] = u;
            }
        };
        auto commit_w = [&](int buf) __attribute__((always_inline)) {
            u32x4* wb = wl + buf * C::W_UNITS;
#pragma unroll
            for (int r = 0; r < C::WR; ++r) {
                const int e = tid + r * C::T;
                if (e < C::W_UNITS) wb[e] = wv[r];
            }
        };
        using I0 = std::integral_constant<int, 0>;
        // Double-buffered path, bf16x3: the three steps of a 16-channel chunk c are (x_hi, w_hi), (x_hi, w_lo), (x_lo, w_hi), and
        // patch and weight buffers are indexed SEPARATELY so that each step stages only the image that changes: x_hi lives in
        // patch buffer 0 for steps 0 and 1, x_lo goes to buffer 1 during step 1; w_hi(c) lives in weight buffer c & 1 for steps
        // 0 and 2, w_lo(c) goes to the other one during step 0; x_hi / w_hi of chunk c+1 are written during step 2 (into patch
        // buffer 0 and weight buffer (c+1) & 1, both last read in step 1).  Four images staged per chunk instead of six.
        auto need_x = [&](int step) __attribute__((always_inline)) { return !(C::DB && PASSES == 3) || (step % 3) != 1; };
        auto need_w = [&](int step) __attribute__((always_inline)) { return !(C::DB && PASSES == 3) || (step % 3) != 2; };
        auto xbuf_of = [&](int step) __attribute__((always_inline)) { return !C::DB ? 0 : PASSES == 3 ? ((step % 3) == 2 ? 1 : 0) : (step & 1); };
        auto wbuf_of = [&](int step) __attribute__((always_inline)) {
            if (!C::DB) return C::WDB ? (step & 1) : 0;
            if (PASSES != 3) return step & 1;
            const int c = step / 3, ps = step - 3 * c;
            return ps == 1 ? 1 - (c & 1) : (c & 1);
        };
        // everything a step needs in flight at once (one workgroup per CU; also the prologue of an item)
        auto issue_all = [&](int step) __attribute__((always_inline)) {
            if (need_x(step)) {
                pp_steps<0, XRL>([&](auto R) __attribute__((always_inline)) { issue_x(step, R, R); });
                load_aff(step);
            }
            if (need_w(step)) issue_w(step);
        };
        auto commit_all = [&](int step) __attribute__((always_inline)) {
            if (need_x(step)) {
                const int xb_ = xbuf_of(step);
                pp_steps<0, XRL>([&](auto R) __attribute__((always_inline)) { commit_x(step, xb_, R, R); });
            }
            if (need_w(step)) commit_w(wbuf_of(step));
        };
        // single-buffered: rounds 1.. of the patch and the weight image, fetched and written one after the other
        auto stage_rest = [&](int step) __attribute__((always_inline)) {
            pp_steps<1, C::XR>([&](auto R) __attribute__((always_inline)) { issue_x(step, R, I0{}); commit_x(step, 0, R, I0{}); });
            if constexpr (!C::WDB) {
                issue_w(step);
                commit_w(0);
            }
        };

        if (any) {
            C16_T(6)
            issue_all(0);
            commit_all(0);
            if constexpr (!C::DB) stage_rest(0);
            else if (nsteps > 1 && !(p.dbg & 1)) issue_all(1);
            __syncthreads();
            C16_T(0)
            for (int step = 0; step < nsteps; ++step) {
                const bool stage = step + 1 < nsteps && !(p.dbg & 1);
                if (stage) {
                    if constexpr (C::DB) {
                        // ONE register set, written AFTER the barrier and re-requested at once: the data of step s+1 (requested a
                        // whole step ago) goes to the other buffer -- free since the barrier that closed step s-1 -- and the
                        // requests of step s+2 leave right behind it, so loads are in flight during every phase of the step
                        // (the first version requested at the top of a step and wrote at its end: one burst per step, and the
                        // memory pipe idle while the wave multiplied and converted)
                        if (!(p.dbg & 128)) commit_all(step + 1);
                        C16_T(4)
                        if (step + 2 < nsteps) issue_all(step + 2);
                    } else {
                        issue_x(step + 1, I0{}, I0{});
                        load_aff(step + 1);
                        if constexpr (C::WDB) issue_w(step + 1);
                    }
                }
                C16_T(1)
                const u32x4* xb = xl + xbuf_of(step) * C::X_UNITS;
                const u32x4* wb = wl + wbuf_of(step) * C::W_UNITS;
                // operand reads run ONE TAP AHEAD of the MFMAs that consume them (two register sets, order pinned with
                // sched_barrier): left alone, hipcc issues a tap's ds_reads right in front of its MFMAs and every tap eats the LDS latency
                u32x4 a[2][MT], b[2][NT];
                auto load_ops = [&](auto T, int set) __attribute__((always_inline)) {
                    constexpr int tap = decltype(T)::value;
                    constexpr int ky = tap / 3, kx = tap % 3;
                    // stride 1: column px + kx + 3; stride 2: input column 2 px + kx + 3 -> plane (kx+3)&1, index px + (kx+3)/2
                    constexpr int toff = ky * C::RS + (S == 1 ? kx : ((kx + 3) & 1) * C::HALF + ((kx + 3) >> 1));
#pragma unroll
                    for (int i = 0; i < MT; ++i) a[set][i] = wb[tap * 2 * C::BM + aoff + i * 32];
#pragma unroll
                    for (int j = 0; j < NT; ++j) b[set][j] = xb[boff[j] + toff];
                };
                load_ops(std::integral_constant<int, 0>{}, 0);
                pp_steps<0, 9>([&](auto T) __attribute__((always_inline)) {
                    constexpr int tap = decltype(T)::value;
                    constexpr int cur = tap & 1;
                    if constexpr (tap + 1 < 9) load_ops(std::integral_constant<int, tap + 1>{}, cur ^ 1);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int i = 0; i < MT; ++i)
#pragma unroll
                        for (int j = 0; j < NT; ++j) acc[i][j] = mfma16<PREC>(a[cur][i], b[cur][j], acc[i][j]);
                    __builtin_amdgcn_sched_barrier(0);
                });
                C16_T(2)
                if constexpr (C::DB) {
                    __syncthreads();
                    C16_T(5)
                } else {
                    if constexpr (C::WDB)
                        if (stage && !(p.dbg & 128)) commit_w(wbuf_of(step + 1)); // the other image: nobody reads it in this step
                    __syncthreads(); // every wave is done reading the operands of this step
                    C16_T(3)
                    if (stage) {
                        if (!(p.dbg & 128)) { commit_x(step + 1, 0, I0{}, I0{}); stage_rest(step + 1); }
                        C16_T(4)
                        __syncthreads();
                        C16_T(5)
                    }
                }
            }
        }

        // ---- epilogue: residual, stores, statistics ----
        // The MFMA leaves a lane with 16 channels of ONE pixel: stored as it stands that is 160 dword stores (and 160 dword
        // residual loads) per lane, and the CU's memory pipe issues those at ~4 B/clk -- the first version of this kernel spent
        // 80 % of its time there.  Each 32 x 32 block goes through a wave-private LDS tile instead ([channel][36]: 16
        // ds_write_b32, 4 ds_read_b128) and comes back as 4 consecutive pixels of one channel per lane: 4 dwordx4 per block, a
        // wave instruction = 8 channels x 128 contiguous bytes.  The residual quads of block b + 2 are requested while block b
        // is transposed (the compiler must assume residual and output alias, so nothing is left for it to hoist).
        if (p.dbg & 4) { if (acc[0][0][0] == 123.456f) p.out[0] = 1.f; if (!C::DB) __syncthreads(); continue; }
        float* gout = OUT16 ? reinterpret_cast<float*>(reinterpret_cast<_Float16*>(p.out) + (size_t)fr * p.out_fs) : p.out + (size_t)fr * p.out_fs;
        const float* gres = !p.res ? nullptr
                            : OUT16 ? reinterpret_cast<const float*>(reinterpret_cast<const _Float16*>(p.res) + (size_t)fr * p.res_fs) : p.res + (size_t)fr * p.res_fs;
        // a frame's output is < 2 GB (checked on the host): lanes with nothing to store sit 2 GB out, where the descriptor drops them
        constexpr unsigned FAR = 0x80000000u;
        __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(gout, 0, 0x7FFFFFFF, 0x00020000);
        __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(gres ? gres : gout), 0, 0x7FFFFFFF, 0x00020000);
        float* scr = scr_all + wave * (32 * 36);
        const int tq = lane & 7, tc = lane >> 3; // after the transpose: pixel quad and channel-in-octet of this lane
        unsigned qoff[NT];
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int pp = (wn * NT + j) * 32 + 4 * tq; // PW % 4 == 0: the quad lies in one tile row
            const int py = pp / PW, px = pp - py * PW;
            const int oy = oy0 + py, ox = ox0 + px;
            const bool ok = oy < p.Hout && ox < p.Wout; // Wout % 4 == 0: inside or outside as a whole
            qoff[j] = ok ? (unsigned)(((size_t)(co0 + wm * MT * 32 + tc) * out_plane + (size_t)oy * p.Wout + ox) * EBO) : FAR;
        }
        const unsigned plane8 = (unsigned)(out_plane * 8 * EBO); // bytes between channel octets
        constexpr int NB = MT * NT;
        // residual quads requested RD blocks ahead (two workgroups per CU: at the block itself, the partner covers the wait -- or one
        // block ahead when the residual is fp16: kept as loaded, a ring slot is 8 registers instead of 16)
        constexpr int RD = C::DB ? 2 : OUT16 ? 1 : 0;
        f32x4 rres[RD + 1][4];
        auto req_res = [&](auto B) __attribute__((always_inline)) {
            constexpr int b_ = decltype(B)::value;
            if constexpr (b_ < NB) {
                constexpr int i = b_ / NT, j = b_ % NT;
                if (gres) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        if constexpr (OUT16) {
                            const f32x2 h_ = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rr, qoff[j], (unsigned)(i * 4 + k) * plane8, 0));
                            rres[b_ % (RD + 1)][k][0] = h_[0]; // widened where it is added
                            rres[b_ % (RD + 1)][k][1] = h_[1];
                        } else
                            rres[b_ % (RD + 1)][k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rr, qoff[j], (unsigned)(i * 4 + k) * plane8, 0));
                    }
                }
            }
        };
        float ps[MT][4], pq[MT][4];
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int k = 0; k < 4; ++k) { ps[i][k] = 0.f; pq[i][k] = 0.f; }
        pp_steps<0, RD>([&](auto B) __attribute__((always_inline)) { req_res(B); });
        pp_steps<0, NB>([&](auto B) __attribute__((always_inline)) {
            constexpr int b_ = decltype(B)::value;
            constexpr int i = b_ / NT, j = b_ % NT;
            req_res(std::integral_constant<int, b_ + RD>{});
#pragma unroll
            for (int e = 0; e < 16; ++e) scr[(8 * (e >> 2) + 4 * kh + (e & 3)) * 36 + n32] = acc[i][j][e];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                f32x4 v = *reinterpret_cast<const f32x4*>(scr + (8 * k + tc) * 36 + 4 * tq);
                if (gres) {
                    if constexpr (OUT16) {
                        const f16x4 h_ = __builtin_bit_cast(f16x4, (f32x2){rres[b_ % (RD + 1)][k][0], rres[b_ % (RD + 1)][k][1]});
                        v += (f32x4){(float)h_[0], (float)h_[1], (float)h_[2], (float)h_[3]};
                    } else
                        v += rres[b_ % (RD + 1)][k];
                }
                if constexpr (OUT16) {
                    const u32x2 h_ = {pack2<P16_FP16>(v[0], v[1]), pack2<P16_FP16>(v[2], v[3])};
                    __builtin_amdgcn_raw_buffer_store_b64(h_, ro, qoff[j], (unsigned)(i * 4 + k) * plane8, 0);
                } else
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), ro, qoff[j], (unsigned)(i * 4 + k) * plane8, 0);
                const float live = (qoff[j] != FAR) ? 1.f : 0.f;
                ps[i][k] += live * ((v[0] + v[1]) + (v[2] + v[3]));
                pq[i][k] += live * ((v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]));
            }
        });
        if (p.stat_acc) {
            // a channel's 8 quad lanes (lane & 7) are the low 3 bits of a DPP row: two quad permutes + row_half_mirror; then over the WN waves
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    float s = ps[i][k], q = pq[i][k];
                    s += dpp_f32<0xB1>(s); q += dpp_f32<0xB1>(q);
                    s += dpp_f32<0x4E>(s); q += dpp_f32<0x4E>(q);
                    s += dpp_f32<0x141>(s); q += dpp_f32<0x141>(q);
                    if (tq == 0) {
                        const int lr = wm * MT * 32 + i * 32 + 8 * k + tc;
                        red[(wn * C::BM + lr) * 2] = s;
                        red[(wn * C::BM + lr) * 2 + 1] = q;
                    }
                }
            __syncthreads();
            double* gstat = p.stat_acc + (size_t)fr * p.stat_fs;
            for (int lr = tid; lr < C::BM; lr += C::T) {
                double s = 0.0, q = 0.0;
#pragma unroll
                for (int w = 0; w < WN; ++w) {
                    s += (double)red[(w * C::BM + lr) * 2];
                    q += (double)red[(w * C::BM + lr) * 2 + 1];
                }
                double* dst = gstat + ((size_t)(item % NREP) * p.stat_C + co0 + lr) * 2;
                atomicAdd(dst, s);
                atomicAdd(dst + 1, q);
            }
        }
        if (p.stat_acc || !C::DB) __syncthreads(); // red (and, single-buffered, the transpose tiles inside the operand buffers) are rewritten by the next item
#if C16_STAMP
        st_[7] += 1;
#endif
    }
#if C16_STAMP
    { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); st_[6] += t_ - tl_; }
    if (p.dbg_buf && tid == 0)
        for (int k = 0; k < 8; ++k) atomicAdd(p.dbg_buf + k, st_[k]);
#endif
}

template <int STRIDE, int PREC, int WM, int WN, int MT, int NT, int PW, int PH, int OCC, int IO16 = 0>
Variant make_c16()
{
    using C = C16<STRIDE, WM, WN, MT, NT, PW, PH, OCC>;
    Variant v;
    v.kern = conv16<STRIDE, PREC, WM, WN, MT, NT, PW, PH, OCC, false, IO16>;
    if constexpr (STRIDE == 2 && !(IO16 & 1)) v.kern2 = conv16<STRIDE, PREC, WM, WN, MT, NT, PW, PH, OCC, true, IO16>; // the sparse first conv reads fp32 PFN rows
    v.io16 = IO16;
    v.bm = C::BM; v.bmp = C::BM; v.pw = PW; v.ph = PH; v.kc = 16; v.threads = C::T;
    v.waves = 4 * OCC; v.pairs = MT * NT; // waves: per CU (the launcher sizes the persistent grid: waves * 64 / threads workgroups per CU)
    v.lds = C::LDS_BYTES;
    v.wino = 5;
    v.prec = PREC;
    if (IO16) snprintf(v.name, sizeof(v.name), "c16 s%d p%d h%d w%dx%d t%dx%d %dx%d o%d", STRIDE, PREC, IO16, WM, WN, MT, NT, PW, PH, OCC);
    else snprintf(v.name, sizeof(v.name), "c16 s%d p%d w%dx%d t%dx%d %dx%d o%d", STRIDE, PREC, WM, WN, MT, NT, PW, PH, OCC);
    return v;
}

template <int STRIDE, int PREC, int WM, int WN, int MT, int NT, int PW, int PH, int OCC, int IO16 = 0>
void add_c16(std::vector<Variant>& m)
{
    using C = C16<STRIDE, WM, WN, MT, NT, PW, PH, OCC>;
    if constexpr (C::LDS_BYTES * C::WGS_PER_CU <= 160 * 1024) m.push_back(make_c16<STRIDE, PREC, WM, WN, MT, NT, PW, PH, OCC, IO16>());
}

// Tile menu.  A tile is PW x PH output pixels = whole 32-pixel N-tiles in row-major order, so PW only has to divide the map:
// eight_20cm's 400 / 200 / 100 maps take 80 / 40 / 20, nuScenes' 240 / 120 / 60 too.  Which entry runs a layer is measured
// on the device (autotune, conv.hip).  Shapes whose LDS image does not fit are dropped at compile time (stride 2 patches
// are 4x the output tile).  OCC 2 = two workgroups per CU (<= 256 registers, single-buffered LDS): the loads, the
// normalise-and-round staging and the store tail of one workgroup run under the MFMAs of the other -- what the layers
// that are bound by the memory pipe need; OCC 1 = one workgroup with the whole register file and double-buffered LDS.
template <int STRIDE, int PREC, int IO16 = 0>
void menu_for(std::vector<Variant>& m)
{
    //              WM WN MT NT  PW  PH OCC    rows  pixels
    if constexpr (STRIDE == 1) {
        add_c16<1, PREC, 2, 2, 1, 5, 40, 8, 2, IO16>(m);   //  64   320
        add_c16<1, PREC, 2, 2, 1, 5, 80, 4, 2, IO16>(m);   //  64   320
        add_c16<1, PREC, 2, 2, 1, 5, 20, 16, 2, IO16>(m);  //  64   320
        add_c16<1, PREC, 4, 1, 1, 5, 20, 8, 2, IO16>(m);   // 128   160
        add_c16<1, PREC, 4, 1, 1, 5, 40, 4, 2, IO16>(m);   // 128   160
        add_c16<1, PREC, 4, 2, 1, 5, 40, 8, 2, IO16>(m);   // 128   320   eight waves in ONE workgroup: one weight image and one patch per 320 / 640
        add_c16<1, PREC, 4, 2, 1, 5, 20, 16, 2, IO16>(m);  // 128   320   pixels instead of one per 160 / 320 (the 128- / 256-channel layers are bound by
        add_c16<1, PREC, 2, 4, 1, 5, 80, 8, 2, IO16>(m);   //  64   640   L2 -> CU bytes per flop, profiles/r03_conv16_stamps.txt)
        add_c16<1, PREC, 1, 4, 2, 5, 80, 8, 1, IO16>(m);   //  64   640
        add_c16<1, PREC, 2, 2, 2, 5, 40, 8, 1, IO16>(m);   // 128   320
        add_c16<1, PREC, 2, 2, 2, 5, 20, 16, 1, IO16>(m);  // 128   320
    } else {
        add_c16<2, PREC, 2, 2, 1, 5, 40, 8, 2, IO16>(m);   //  64   320
        add_c16<2, PREC, 2, 2, 1, 5, 20, 16, 2, IO16>(m);  //  64   320
        add_c16<2, PREC, 4, 1, 1, 5, 20, 8, 2, IO16>(m);   // 128   160
        add_c16<2, PREC, 4, 2, 1, 5, 40, 8, 2, IO16>(m);   // 128   320   eight waves
        add_c16<2, PREC, 2, 2, 1, 5, 40, 8, 1, IO16>(m);   //  64   320
        add_c16<2, PREC, 4, 1, 1, 5, 20, 8, 1, IO16>(m);   // 128   160
    }
}

} // namespace

void conv16_menu(int stride, int prec, std::vector<Variant>& menu, int io16)
{
    if (io16) { // pp_set_precision 4: fp16 operands with fp16 tensors (io16 3), the first conv with an fp16 output only (io16 2)
        if (stride == 1) menu_for<1, P16_FP16, 3>(menu);
        else if (io16 == 2) menu_for<2, P16_FP16, 2>(menu);
        else menu_for<2, P16_FP16, 3>(menu);
        return;
    }
    if (stride == 1) {
        if (prec == P16_BF16X3) menu_for<1, P16_BF16X3>(menu);
        else if (prec == P16_BF16) menu_for<1, P16_BF16>(menu);
        else if (prec == P16_FP16) menu_for<1, P16_FP16>(menu);
    } else {
        if (prec == P16_BF16X3) menu_for<2, P16_BF16X3>(menu);
        else if (prec == P16_BF16) menu_for<2, P16_BF16>(menu);
        else if (prec == P16_FP16) menu_for<2, P16_FP16>(menu);
    }
}

} // namespace ppc
