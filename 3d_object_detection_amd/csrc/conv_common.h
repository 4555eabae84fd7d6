// Shared by the conv kernels' translation units (conv.hip: fp32 MFMA family; conv16.hip: 16-bit operand family): kernel
// parameter block, launch-plan entry, XCD-aware block numbering and the small device helpers.  gfx950 only.
#pragma once
#include <type_traits>
#include "pp_common.h"

namespace ppc {

// Workgroups of a launch are dealt round-robin over the 8 XCDs in linear-id order (x fastest), each XCD with
// its own 4 MB L2.  Re-number them so that XCD k works through the k-th CONTIGUOUS eighth of the
// (cout-block fastest, then tile, then frame) order: the cout blocks of one tile (same input patch) and
// neighbouring tiles (shared halo lines) then meet in one L2 instead of each fetching across the fabric.
struct BlockId { int x, y, z; };
static __device__ __forceinline__ BlockId xcd_block_id()
{
    const unsigned gx = gridDim.x, gy = gridDim.y, gz = gridDim.z;
    const unsigned n = gx * gy * gz;
    const unsigned lin = blockIdx.x + gx * (blockIdx.y + gy * blockIdx.z);
    const unsigned per = n >> 3;
    const unsigned l2 = (lin < per * 8) ? (lin & 7) * per + (lin >> 3) : lin;
    BlockId b;
    b.y = (int)(l2 % gy);
    const unsigned t = l2 / gy;
    b.x = (int)(t % gx);
    b.z = (int)(t / gx);
    return b;
}

typedef float f32x4 __attribute__((ext_vector_type(4)));
// dword-aligned vector stores (global memory takes multi-dword accesses at dword alignment)
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
typedef float f32x2u __attribute__((ext_vector_type(2), aligned(4)));

// Sum over the 16 lanes that share lane >> 4 (one row of the 16x16 MFMA tile = one DPP row), on the VALU: two quad
// permutes, row_half_mirror, row_mirror.  After each step all lanes of the merged group hold the same value, so the
// result is bit-identical to the xor-shuffle butterfly (1, 2, 4, 8) it replaces -- which hipcc turned into four
// dependent ds_bpermute per value (64 LDS round trips per Winograd tile and wave).
template <int CTRL>
static __device__ __forceinline__ float dpp_f32(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
static __device__ __forceinline__ float row16_sum(float v)
{
    v += dpp_f32<0xB1>(v);  // quad_perm [1,0,3,2]
    v += dpp_f32<0x4E>(v);  // quad_perm [2,3,0,1]
    v += dpp_f32<0x141>(v); // row_half_mirror
    v += dpp_f32<0x140>(v); // row_mirror
    return v;
}

constexpr int NREP = 8; // replicated statistics accumulators (spreads atomic contention)

// compile-time for: f(integral_constant<int, I>) for I in [B, E)
template <int B, int E, typename F>
static __device__ __forceinline__ void pp_steps(F&& f)
{
    if constexpr (B < E) {
        f(std::integral_constant<int, B>{});
        pp_steps<B + 1, E>(f);
    }
}

enum { EPI_PLAIN = 0, EPI_UP2 = 1, EPI_UP4 = 2, EPI_HEAD = 3 };
enum { PRE_RAW = 0, PRE_STATS = 1, PRE_AFFINE = 2 };

struct ConvP {
    const float* in;
    const float* w;   // packed [cout_block][chunk][tap][kc][BM]
    float* out;
    const float* res; // residual, same layout as out (nullable)
    int Cin, Hin, Win;
    int Cout;         // rows of the GEMM (virtual channels for deconv, 96 for the head)
    int Hout, Wout;   // pixel grid of the GEMM
    int pre;
    const double* pre_acc; // [NREP][Cin][2]
    const float* pre_scale;
    const float* pre_shift;
    double pre_inv_n;
    float eps;
    double* stat_acc; // [NREP][Cstat][2] (nullable)
    int stat_C;       // channels in stat_acc
    // head
    const float* bias;
    float* out_box;
    float* out_dir;
    int n_cls, n_box; // na, 7 na (dir = rest up to n_rows) for na anchors per location (reference: 9, 63)
    int n_rows;       // 10 na (reference: 90)
    int dbg;          // diagnostics only (PP_CONV_DBG): 1 = skip staging after chunk 0, 4 = skip epilogue
    // batch: blockIdx.z = frame; strides in elements between consecutive frames
    size_t in_fs, out_fs, res_fs, box_fs, dir_fs; // floats
    size_t pre_fs, stat_fs;                        // doubles
    size_t aff_fs;                                 // floats between frames of pre_scale / pre_shift (0: shared)
    unsigned long long* dbg_buf;                   // diagnostic builds only (PP_WINO_STAMP): stamp sums
    int nb;                                        // frames (persistent kernels loop over them; others use grid.z)
    // sparse BEV input of the first conv: pillar-index map [Hin*Win] (-1 = empty) + PFN rows [P][64]
    const int32_t* pmap;
    const float* feat;
    size_t pmap_fs, feat_fs;
    // wino4_mfma: the rectangle of output pixels this launch tiles (a layer whose map is not a multiple of the tile is
    // covered by a main launch of whole tiles plus strip launches of thin tiles): origin, exclusive end, tiles in x / y
    int rx0, ry0, rx1, ry1, rnbx, rnby;
};

struct Variant { // one compiled tiling of conv_mfma
    void (*kern)(const ConvP);
    void (*kern2)(const ConvP) = nullptr; // conv16, stride 2: twin for the sparse BEV input of the first conv (ConvP::pmap)
    int bm, bmp, pw, ph, kc, threads, waves, pairs; // pairs = MT*NT tile pairs per wave
    size_t lds;
    char name[48];
    int wino = 0; // 1: Winograd F(2x2,3x3) image; 2: slab-resident persistent Winograd; 3: persistent 1x1 GEMM (weights resident); 4: wino4_mfma;
                  // 5: conv16; 6: Winograd F(4x4,3x3), positions split over the waves (wino6.hip)
    int cin = 0;  // wino == 2: compiled for exactly this Cin
    int prec = 0; // wino == 3 / 5: 0 fp32 MFMA, 1 split-bf16 (bf16x3), 2 plain bf16, 3 fp16 operands
    int io16 = 0; // wino == 3 / 5: bit 0 = the input tensor is fp16, bit 1 = the output tensor (and residual) is (pp_set_precision 4)
};


// conv16.hip: 16-bit operand 3x3 convolutions (fp16 / bf16 / split-bf16) -- menu entries for one layer shape and precision
void conv16_menu(int stride, int prec, std::vector<Variant>& menu, int io16 = 0);

// wino6.hip: Winograd F(4x4,3x3) on fp32 MFMA -- menu entry, the strip tilings of its region launches, the weight image
void wino6_menu(std::vector<Variant>& menu, bool roofline_layer);
Variant wino6_strip_v(); // 4 px wide, 64 px tall
Variant wino6_strip_h(); // 64 px wide, 4 px tall
void wino6_pack(const float* w /*[rows][cin][3][3]*/, int rows, int cin, std::vector<float>& out);
constexpr int W6_FRONT_PAD = 64; // floats in front of every tensor a wino6 launch reads: its dwordx4 patch pieces start one float before a row

} // namespace ppc
