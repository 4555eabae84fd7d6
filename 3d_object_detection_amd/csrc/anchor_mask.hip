// Per-frame occupied-anchor mask (framework/anchor_assigner.py:322-335; the reference's CUDA
// path is four numba kernels, box_np_ops.py:168-257, with serial per-column scans).
//   M1  occupancy map from pillar coords (int32; counts < 2^24 so the reference's f32 is exact too)
//   M2  inclusive scan along y (contiguous): one wave per row, 64-wide chunks with carry
//   M3  inclusive scan along x: one 1024-thread block per 64-column strip, rows split over
//       the 16 waves (partial sums through LDS, then a second sweep)
//   M4  4-tap summed-area lookup per anchor, no -1 offsets (box_np_ops.py:217-227 quirk kept).
//       The clamped cell rectangle of an anchor factors into (type, ix) and (type, iy) tables
//       of a few KB, so the lookup reads no per-anchor table from HBM; a full [A,4] table is
//       kept as fallback if the host finds the rectangles not separable.
#include "pp_common.h"

namespace {

__device__ __forceinline__ void occ_mark_body(const int32_t* __restrict__ coors, const int32_t* __restrict__ num_pillars,
                                                int gy, int32_t* __restrict__ occ)
{
    int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= *num_pillars) return;
    atomicAdd(&occ[coors[3 * p] * gy + coors[3 * p + 1]], 1);
}

__device__ __forceinline__ void scan_rows_body(int32_t* __restrict__ occ, int gx, int gy)
{
    int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    int lane = threadIdx.x & 63;
    if (row >= gx) return;
    int32_t* r = occ + (size_t)row * gy;
    int carry = 0;
    constexpr int NB = 8; // 64-cell pieces requested together: the carry chain then runs on registers, not through one memory round trip per piece
    for (int yb = 0; yb < gy; yb += 64 * NB) {
        int vv[NB];
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const int y = yb + b * 64 + lane;
            const int x = r[y < gy ? y : gy - 1];
            vv[b] = y < gy ? x : 0;
        }
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const int y = yb + b * 64 + lane;
            int v = vv[b];
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                int t = __shfl_up(v, o);
                if (lane >= o) v += t;
            }
            v += carry;
            if (y < gy) r[y] = v;
            carry = __shfl(v, 63);
        }
    }
}

__device__ __forceinline__ void scan_cols_body(int32_t* __restrict__ occ, int gx, int gy)
{
    __shared__ int part[16][64];
    int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int col = blockIdx.x * 64 + lane;
    int rows_per = (gx + 15) / 16;
    int x0 = w * rows_per, x1 = min(gx, x0 + rows_per);
    // both passes in batches of CH rows requested together (clamped addresses, selects): row by row the segment was a chain of
    // ~2 x 50 memory round trips on the 13 workgroups a 800-cell grid gives this kernel (23 us of a one-frame call)
    constexpr int CH = 16;
    int s = 0;
    if (col < gy)
        for (int x = x0; x < x1; x += CH) {
            int v[CH];
#pragma unroll
            for (int i = 0; i < CH; ++i) v[i] = occ[(size_t)min(x + i, x1 - 1) * gy + col];
#pragma unroll
            for (int i = 0; i < CH; ++i) s += (x + i < x1) ? v[i] : 0;
        }
    part[w][lane] = s;
    __syncthreads();
    int run = 0;
    for (int k = 0; k < w; ++k) run += part[k][lane];
    if (col < gy)
        for (int x = x0; x < x1; x += CH) {
            int v[CH];
#pragma unroll
            for (int i = 0; i < CH; ++i) v[i] = occ[(size_t)min(x + i, x1 - 1) * gy + col];
#pragma unroll
            for (int i = 0; i < CH; ++i)
                if (x + i < x1) {
                    run += v[i];
                    occ[(size_t)(x + i) * gy + col] = run;
                }
        }
}

__device__ __forceinline__ void mask_lookup_sep_body(const int32_t* __restrict__ sat, int gy, int H, int W, int types,
                                                       const int32_t* __restrict__ rect_x, const int32_t* __restrict__ rect_y,
                                                       uint8_t* __restrict__ mask)
{
    // anchor index a = (type * H + ix) * W + iy
    int64_t a = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t total = (int64_t)types * H * W;
    if (a >= total) return;
    int iy = (int)(a % W);
    int64_t q = a / W;
    int ix = (int)(q % H);
    int t = (int)(q / H);
    int minx = rect_x[(t * H + ix) * 2], maxx = rect_x[(t * H + ix) * 2 + 1];
    int miny = rect_y[(t * W + iy) * 2], maxy = rect_y[(t * W + iy) * 2 + 1];
    int area = sat[maxx * gy + maxy] - sat[maxx * gy + miny] - sat[minx * gy + maxy] + sat[minx * gy + miny];
    mask[a] = area > 0;
}

__device__ __forceinline__ void mask_lookup_full_body(const int32_t* __restrict__ sat, int gy, int64_t A,
                                                        const int4* __restrict__ rects, uint8_t* __restrict__ mask)
{
    int64_t a = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= A) return;
    int4 r = rects[a]; // minx, miny, maxx, maxy
    int area = sat[r.z * gy + r.w] - sat[r.z * gy + r.y] - sat[r.x * gy + r.w] + sat[r.x * gy + r.y];
    mask[a] = area > 0;
}

__global__ void __launch_bounds__(256) occ_mark(const int32_t* __restrict__ coors, const int32_t* __restrict__ num_pillars, int gy, int32_t* __restrict__ occ)
{
    occ_mark_body(coors, num_pillars, gy, occ);
}
__global__ void __launch_bounds__(256) scan_rows(int32_t* __restrict__ occ, int gx, int gy) { scan_rows_body(occ, gx, gy); }
__global__ void __launch_bounds__(1024) scan_cols(int32_t* __restrict__ occ, int gx, int gy) { scan_cols_body(occ, gx, gy); }
__global__ void __launch_bounds__(256) mask_lookup_sep(const int32_t* __restrict__ sat, int gy, int H, int W, int types,
                                                       const int32_t* __restrict__ rect_x, const int32_t* __restrict__ rect_y, uint8_t* __restrict__ mask)
{
    mask_lookup_sep_body(sat, gy, H, W, types, rect_x, rect_y, mask);
}
__global__ void __launch_bounds__(256) mask_lookup_full(const int32_t* __restrict__ sat, int gy, int64_t A, const int4* __restrict__ rects, uint8_t* __restrict__ mask)
{
    mask_lookup_full_body(sat, gy, A, rects, mask);
}
// batched twins: blockIdx.z = frame
__global__ void __launch_bounds__(256) occ_mark_b(const pp_pre_frame* __restrict__ tab, int gy)
{
    const pp_pre_frame F = tab[blockIdx.z];
    occ_mark_body(F.coors, F.num, gy, F.occ);
}
__global__ void __launch_bounds__(256) scan_rows_b(const pp_pre_frame* __restrict__ tab, int gx, int gy) { scan_rows_body(tab[blockIdx.z].occ, gx, gy); }
__global__ void __launch_bounds__(1024) scan_cols_b(const pp_pre_frame* __restrict__ tab, int gx, int gy) { scan_cols_body(tab[blockIdx.z].occ, gx, gy); }
__global__ void __launch_bounds__(256) mask_lookup_sep_b(const pp_pre_frame* __restrict__ tab, int gy, int H, int W, int types,
                                                         const int32_t* __restrict__ rect_x, const int32_t* __restrict__ rect_y)
{
    const pp_pre_frame F = tab[blockIdx.z];
    mask_lookup_sep_body(F.occ, gy, H, W, types, rect_x, rect_y, F.mask);
}
__global__ void __launch_bounds__(256) mask_lookup_full_b(const pp_pre_frame* __restrict__ tab, int gy, int64_t A, const int4* __restrict__ rects)
{
    const pp_pre_frame F = tab[blockIdx.z];
    mask_lookup_full_body(F.occ, gy, A, rects, F.mask);
}

} // namespace

extern "C" int pp_anchor_mask(pp_ctx* ctx, const int32_t* coors, const int32_t* num_pillars, uint8_t* mask, void* stream_)
{
    if (!ctx) return PP_E_ARG;
    return pp_anchor_mask_slot(ctx, 0, coors, num_pillars, mask, (hipStream_t)stream_);
}

int pp_anchor_mask_slot(pp_ctx* ctx, int si, const int32_t* coors, const int32_t* num_pillars, uint8_t* mask, hipStream_t stream)
{
    pp_slot& S = ctx->slot[si];
    if (!coors || !num_pillars || !mask) return pp_fail(ctx, PP_E_ARG, "pp_anchor_mask: null pointer");
    if (ctx->A == 0) return pp_fail(ctx, PP_E_STATE, "pp_anchor_mask: call pp_set_anchors first");
    const int gx = ctx->gx, gy = ctx->gy;
    PP_HIP(hipMemsetAsync(S.occ, 0, (size_t)gx * gy * sizeof(int32_t), stream));
    hipLaunchKernelGGL(occ_mark, dim3(pp_div_up(ctx->cfg.max_voxels, 256)), dim3(256), 0, stream, coors, num_pillars, gy, S.occ);
    hipLaunchKernelGGL(scan_rows, dim3(pp_div_up(gx, 4)), dim3(256), 0, stream, S.occ, gx, gy);
    hipLaunchKernelGGL(scan_cols, dim3(pp_div_up(gy, 64)), dim3(1024), 0, stream, S.occ, gx, gy);
    if (ctx->rect_separable) {
        int types = (int)(ctx->A / ((int64_t)ctx->H * ctx->W));
        hipLaunchKernelGGL(mask_lookup_sep, dim3(pp_div_up(ctx->A, 256)), dim3(256), 0, stream, S.occ, gy, ctx->H, ctx->W,
                           types, ctx->rect_x, ctx->rect_y, mask);
    } else {
        hipLaunchKernelGGL(mask_lookup_full, dim3(pp_div_up(ctx->A, 256)), dim3(256), 0, stream, S.occ, gy, ctx->A,
                           (const int4*)ctx->rects, mask);
    }
    PP_HIP(hipGetLastError());
    return 0;
}

// occ tables were zeroed by pre_init_b (voxelize.hip)
int pp_anchor_mask_group(pp_ctx* ctx, int b0, int g, hipStream_t stream)
{
    const pp_pre_frame* tab = ctx->d_pre + b0;
    const int gx = ctx->gx, gy = ctx->gy;
    hipLaunchKernelGGL(occ_mark_b, dim3(pp_div_up(ctx->cfg.max_voxels, 256), 1, g), dim3(256), 0, stream, tab, gy);
    hipLaunchKernelGGL(scan_rows_b, dim3(pp_div_up(gx, 4), 1, g), dim3(256), 0, stream, tab, gx, gy);
    hipLaunchKernelGGL(scan_cols_b, dim3(pp_div_up(gy, 64), 1, g), dim3(1024), 0, stream, tab, gx, gy);
    if (ctx->rect_separable) {
        int types = (int)(ctx->A / ((int64_t)ctx->H * ctx->W));
        hipLaunchKernelGGL(mask_lookup_sep_b, dim3(pp_div_up(ctx->A, 256), 1, g), dim3(256), 0, stream, tab, gy, ctx->H, ctx->W, types,
                           ctx->rect_x, ctx->rect_y);
    } else {
        hipLaunchKernelGGL(mask_lookup_full_b, dim3(pp_div_up(ctx->A, 256), 1, g), dim3(256), 0, stream, tab, gy, ctx->A, (const int4*)ctx->rects);
    }
    PP_HIP(hipGetLastError());
    return 0;
}
