// Pillar feature net and BEV scatter.
//
// pfn_kernel (networks/pointpillars8_shared.py:30-60 as ONE kernel instead of ~15 eager ops):
//   one wavefront per pillar, lane = output channel (64 channels == 64 lanes), so the max over
//   the T point slots is a per-lane running max -- no cross-lane reduction, no [P,T,9] temporary.
//   The pillar's T x F point tile is loaded once, coalesced (lane j loads float j), kept in
//   registers and broadcast with readlane; the 3-component mean is a 4-step xor-shuffle reduction.
//   Conv1d(9->64, no bias) + BatchNorm1d(eval) fold into 9 FMAs + one FMA (scale, shift);
//   padded slots (zeroed BEFORE the conv, :47-50) contribute relu(shift[c]) to the max.
//   HBM: 16*T B read + 256 B written per pillar; the 576-float weight lives in VGPRs.
//
// scatter_kernel (:76-111 / CUDA string of pointpillars8_trt.py:176-193): zero fill + scatter
//   into the dense [64, gx, gy] canvas.  Only the stand-alone pp_scatter uses it; the fused
//   frame path feeds the first conv from a pillar-index map instead (conv.hip).
#include "pp_common.h"

namespace {

__device__ __forceinline__ void pfn_kernel_body(const float* __restrict__ voxels, const int32_t* __restrict__ coors,
                                                  const int32_t* __restrict__ npts, const int32_t* __restrict__ num_pillars,
                                                  const float* __restrict__ wT /*[9][64]*/, const float* __restrict__ scale,
                                                  const float* __restrict__ shift, float vx, float vy, float x_off,
                                                  float y_off, int T, float* __restrict__ feat)
{
    const int lane = threadIdx.x & 63;
    const int P = *num_pillars;
    float w[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) w[k] = wT[k * 64 + lane];
    const float sc = scale[lane], sh = shift[lane];
    const float pad = fmaxf(sh, 0.f);
    const int waves = (gridDim.x * blockDim.x) >> 6;
    for (int p = (blockIdx.x * blockDim.x + threadIdx.x) >> 6; p < P; p += waves) {
        const float* v = voxels + (size_t)p * T * 4;
        const int n = min(npts[p], T); // a count beyond the T slots of the pillar buffer would read the next pillar's rows
        // pass 1: sum of x,y,z over ALL T slots (padded slots are zero), as :32 does
        float s = 0.f;
        for (int j = lane; j < T * 4; j += 64) s += v[j];
        // lanes with equal (lane & 3) hold the same component
        s += __shfl_xor(s, 4); s += __shfl_xor(s, 8); s += __shfl_xor(s, 16); s += __shfl_xor(s, 32);
        const float fn = (float)n;
        const float mx = __shfl(s, 0) / fn, my = __shfl(s, 1) / fn, mz = __shfl(s, 2) / fn;
        // separate multiply and add, as torch evaluates coors.float() * vx + x_offset (:37-38)
        const float cxf = __fadd_rn(__fmul_rn((float)coors[3 * p], vx), x_off);
        const float cyf = __fadd_rn(__fmul_rn((float)coors[3 * p + 1], vy), y_off);
        float best = (n < T) ? pad : -INFINITY;
        for (int t0 = 0; t0 < n; t0 += 16) {
            const int j = t0 * 4 + lane;
            const float mine = (j < T * 4) ? v[j] : 0.f;
            const int tn = min(16, n - t0);
            for (int t = 0; t < tn; ++t) {
                const float x = __shfl(mine, 4 * t), y = __shfl(mine, 4 * t + 1), z = __shfl(mine, 4 * t + 2),
                            r = __shfl(mine, 4 * t + 3);
                float a = w[0] * x;
                a = fmaf(w[1], y, a);
                a = fmaf(w[2], z, a);
                a = fmaf(w[3], r, a);
                a = fmaf(w[4], x - mx, a);
                a = fmaf(w[5], y - my, a);
                a = fmaf(w[6], z - mz, a);
                a = fmaf(w[7], x - cxf, a);
                a = fmaf(w[8], y - cyf, a);
                best = fmaxf(best, fmaxf(fmaf(a, sc, sh), 0.f));
            }
        }
        feat[(size_t)p * 64 + lane] = best;
    }
}

__global__ void __launch_bounds__(256) scatter_kernel(const float* __restrict__ feat, const int32_t* __restrict__ coors,
                                                      const int32_t* __restrict__ num_pillars, int gx, int gy, size_t plane,
                                                      float* __restrict__ canvas)
{
    const int lane = threadIdx.x & 63;
    const int P = *num_pillars;
    const int waves = (gridDim.x * blockDim.x) >> 6;
    for (int p = (blockIdx.x * blockDim.x + threadIdx.x) >> 6; p < P; p += waves) {
        const int cx = coors[3 * p], cy = coors[3 * p + 1];
        if ((unsigned)cx >= (unsigned)gx || (unsigned)cy >= (unsigned)gy) continue; // a coordinate outside the grid is skipped, never written
        size_t cell = (size_t)cx * gy + cy;
        canvas[(size_t)lane * plane + cell] = feat[(size_t)p * 64 + lane];
    }
}

__device__ __forceinline__ void pmap_kernel_body(const int32_t* __restrict__ coors, const int32_t* __restrict__ num_pillars, int gx, int gy,
                                                   int32_t* __restrict__ pmap)
{
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p < *num_pillars) {
        const int cx = coors[3 * p], cy = coors[3 * p + 1];
        if ((unsigned)cx < (unsigned)gx && (unsigned)cy < (unsigned)gy) pmap[(size_t)cx * gy + cy] = p;
    }
}

__global__ void __launch_bounds__(256) pfn_kernel(const float* __restrict__ voxels, const int32_t* __restrict__ coors, const int32_t* __restrict__ npts,
                                                  const int32_t* __restrict__ num_pillars, const float* __restrict__ wT, const float* __restrict__ scale,
                                                  const float* __restrict__ shift, float vx, float vy, float x_off, float y_off, int T, float* __restrict__ feat)
{
    pfn_kernel_body(voxels, coors, npts, num_pillars, wT, scale, shift, vx, vy, x_off, y_off, T, feat);
}
__global__ void __launch_bounds__(256) pmap_kernel(const int32_t* __restrict__ coors, const int32_t* __restrict__ num_pillars, int gx, int gy, int32_t* __restrict__ pmap)
{
    pmap_kernel_body(coors, num_pillars, gx, gy, pmap);
}
// batched twin: PFN rows and the pillar map of frame blockIdx.z in one launch (the map was filled with -1 by pre_init_b)
__global__ void __launch_bounds__(256) pfn_pmap_b(const pp_pre_frame* __restrict__ tab, const float* __restrict__ wT, const float* __restrict__ scale,
                                                  const float* __restrict__ shift, float vx, float vy, float x_off, float y_off, int T, int gx, int gy, int pmap_blocks)
{
    const pp_pre_frame F = tab[blockIdx.z];
    pfn_kernel_body(F.voxels, F.coors, F.npts, F.num, wT, scale, shift, vx, vy, x_off, y_off, T, F.feat);
    if ((int)blockIdx.x < pmap_blocks) pmap_kernel_body(F.coors, F.num, gx, gy, F.pmap);
}

} // namespace

// Sparse form of the BEV canvas for the fused path: cell -> pillar id (-1 = empty).  2.56 MB at 800^2 instead of
// a 164 MB zero-filled canvas; the first conv gathers its input rows from the [P,64] PFN output through it.
int pp_pillar_map(pp_ctx* ctx, const int32_t* coors, const int32_t* num_pillars, int32_t* pmap, hipStream_t stream)
{
    PP_HIP(hipMemsetAsync(pmap, 0xFF, (size_t)ctx->gx * ctx->gy * sizeof(int32_t), stream));
    hipLaunchKernelGGL(pmap_kernel, dim3(pp_div_up(ctx->cfg.max_voxels, 256)), dim3(256), 0, stream, coors, num_pillars, ctx->gx, ctx->gy, pmap);
    PP_HIP(hipGetLastError());
    return 0;
}

extern "C" int pp_pfn(pp_ctx* ctx, const float* voxels, const int32_t* coors, const int32_t* npts,
                      const int32_t* num_pillars, float* feat, void* stream_)
{
    if (!ctx) return PP_E_ARG;
    hipStream_t stream = (hipStream_t)stream_;
    if (!ctx->weights_ready) return pp_fail(ctx, PP_E_STATE, "pp_pfn: weights not committed");
    if (!voxels || !coors || !npts || !num_pillars || !feat) return pp_fail(ctx, PP_E_ARG, "pp_pfn: null pointer");
    if (ctx->cfg.num_point_features != 4) return pp_fail(ctx, PP_E_ARG, "pp_pfn: only F=4 point features supported");
    const pp_config& c = ctx->cfg;
    float vx = c.voxel_size[0], vy = c.voxel_size[1];
    float x_off = vx / 2 + c.offset[0], y_off = vy / 2 + c.offset[1]; // :18-19
    hipLaunchKernelGGL(pfn_kernel, dim3(1024), dim3(256), 0, stream, voxels, coors, npts, num_pillars, ctx->pfn_w,
                       ctx->pfn_scale, ctx->pfn_shift, vx, vy, x_off, y_off, c.max_num_points, feat);
    PP_HIP(hipGetLastError());
    return 0;
}

extern "C" int pp_scatter(pp_ctx* ctx, const float* feat, const int32_t* coors, const int32_t* num_pillars,
                          float* canvas, void* stream_)
{
    if (!ctx) return PP_E_ARG;
    hipStream_t stream = (hipStream_t)stream_;
    if (!feat || !coors || !num_pillars || !canvas) return pp_fail(ctx, PP_E_ARG, "pp_scatter: null pointer");
    size_t plane = (size_t)ctx->gx * ctx->gy;
    PP_HIP(hipMemsetAsync(canvas, 0, plane * 64 * sizeof(float), stream));
    hipLaunchKernelGGL(scatter_kernel, dim3(1024), dim3(256), 0, stream, feat, coors, num_pillars, ctx->gx, ctx->gy, plane, canvas);
    PP_HIP(hipGetLastError());
    return 0;
}

int pp_pfn_pmap_group(pp_ctx* ctx, int b0, int g, hipStream_t stream)
{
    if (ctx->cfg.num_point_features != 4) return pp_fail(ctx, PP_E_ARG, "pp_pfn: only F=4 point features supported");
    const pp_config& c = ctx->cfg;
    const float vx = c.voxel_size[0], vy = c.voxel_size[1];
    const float x_off = vx / 2 + c.offset[0], y_off = vy / 2 + c.offset[1]; // :18-19
    const int pmap_blocks = pp_div_up(c.max_voxels, 256);
    const int blocks = pmap_blocks > 256 ? pmap_blocks : 256;
    hipLaunchKernelGGL(pfn_pmap_b, dim3(blocks, 1, g), dim3(256), 0, stream, ctx->d_pre + b0, ctx->pfn_w, ctx->pfn_scale, ctx->pfn_shift, vx, vy,
                       x_off, y_off, c.max_num_points, ctx->gx, ctx->gy, pmap_blocks);
    PP_HIP(hipGetLastError());
    return 0;
}
