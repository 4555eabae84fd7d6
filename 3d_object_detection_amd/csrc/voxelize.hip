// Points -> pillars, bit-exact with the reference's sequential loop
// (framework/voxel_generator.py:82-106) but order-independent in execution:
//
//   K1  cell id per point (IEEE fp32 floorf((p-off)/vs), true division) + atomicMin of the
//       point index into cell_first[cell]              -> first point of every cell
//   K2  first-point flags -> per-wave popcounts (64-wide ballot)
//   K3  rank of each first point = exclusive prefix of the flags = pillar id in order of first
//       appearance; the point that would open pillar #max_voxels is the reference's `break`
//       (:96-97): i* = its index, every point >= i* is dropped
//   K4  ordered slotting without sorting: each point pushes its index through the pillar's T
//       slots with atomicMin, carrying the displaced (larger) index onward.  Whatever the
//       interleaving, slot s ends up holding the (s+1)-th smallest index of the cell, i.e.
//       the first T points in stream order (:103-105)
//   K5  gather points into voxels[P,T,F] (zero padded) and count filled slots
//
// All HBM traffic is the raw cloud (16 B/point) + the [P,T,F] output; cell_first (2.56 MB at
// 800^2) and the slot table live in L2/Infinity Cache.
#include "pp_common.h"

namespace {

__device__ __forceinline__ int cell_of(const float* __restrict__ p, float ox, float oy, float oz, float vx,
                                       float vy, float vz, int gx, int gy, int gz, int& cx, int& cy, int& cz)
{
    // __fdiv_rn / __fsub_rn: no reciprocal, no contraction -- boundaries must match numpy fp32
    float fx = floorf(__fdiv_rn(__fsub_rn(p[0], ox), vx));
    float fy = floorf(__fdiv_rn(__fsub_rn(p[1], oy), vy));
    float fz = floorf(__fdiv_rn(__fsub_rn(p[2], oz), vz));
    bool in = (fx >= 0.f) & (fx < (float)gx) & (fy >= 0.f) & (fy < (float)gy) & (fz >= 0.f) & (fz < (float)gz);
    if (!in) return -1;
    cx = (int)fx; cy = (int)fy; cz = (int)fz;
    return (cx * gy + cy) * gz + cz;
}

__device__ __forceinline__ void vox_cell_first_body(const float* __restrict__ pts, int n, int nfeat, pp_config cfg,
                                                      int32_t* __restrict__ pt_cell, int32_t* __restrict__ cell_first)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int cx, cy, cz;
    int c = cell_of(pts + (size_t)i * nfeat, cfg.offset[0], cfg.offset[1], cfg.offset[2], cfg.voxel_size[0],
                    cfg.voxel_size[1], cfg.voxel_size[2], cfg.grid_size[0], cfg.grid_size[1], cfg.grid_size[2], cx, cy, cz);
    pt_cell[i] = c;
    if (c >= 0) atomicMin(&cell_first[c], i);
}

__device__ __forceinline__ void vox_flag_count_body(const int32_t* __restrict__ pt_cell,
                                                      const int32_t* __restrict__ cell_first, int n,
                                                      int32_t* __restrict__ wave_cnt)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    bool flag = false;
    if (i < n) {
        int c = pt_cell[i];
        flag = (c >= 0) && (cell_first[c] == i);
    }
    unsigned long long b = __ballot(flag);
    if ((threadIdx.x & 63) == 0) wave_cnt[i >> 6] = __popcll(b);
}

// Each wave derives its own exclusive offset by summing the counts of all earlier waves
// (nw <= a few thousand: cheaper than a separate scan launch).
__device__ __forceinline__ void vox_rank_body(const float* __restrict__ pts, int n, int nfeat, pp_config cfg,
                                                const int32_t* __restrict__ pt_cell,
                                                const int32_t* __restrict__ cell_first,
                                                const int32_t* __restrict__ wave_cnt, int32_t* __restrict__ pt_rank,
                                                int32_t* __restrict__ coors, int32_t* __restrict__ scalars,
                                                int32_t* __restrict__ num_pillars)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    int lane = threadIdx.x & 63;
    int w = i >> 6;
    int nw = (n + 63) >> 6;
    int part = 0, tot = 0;
    for (int k = lane; k < nw; k += 64) {
        int v = wave_cnt[k];
        tot += v;
        if (k < w) part += v;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        part += __shfl_xor(part, o);
        tot += __shfl_xor(tot, o);
    }
    bool flag = false;
    int c = -1;
    if (i < n) {
        c = pt_cell[i];
        flag = (c >= 0) && (cell_first[c] == i);
    }
    unsigned long long b = __ballot(flag);
    int rank = part + __popcll(b & ((1ull << lane) - 1ull));
    if (flag) {
        pt_rank[i] = rank;
        if (rank < cfg.max_voxels) {
            int gy = cfg.grid_size[1], gz = cfg.grid_size[2];
            int cz = c % gz, cy = (c / gz) % gy, cx = c / (gz * gy);
            coors[3 * rank + 0] = cx;
            coors[3 * rank + 1] = cy;
            coors[3 * rank + 2] = cz;
        } else if (rank == cfg.max_voxels) {
            scalars[0] = i; // i*: the reference breaks here
        }
    }
    if (i == 0) {
        scalars[1] = tot;
        *num_pillars = tot < cfg.max_voxels ? tot : cfg.max_voxels;
    }
}

__device__ __forceinline__ void vox_insert_body(int n, pp_config cfg, const int32_t* __restrict__ pt_cell,
                                                  const int32_t* __restrict__ cell_first,
                                                  const int32_t* __restrict__ pt_rank,
                                                  const int32_t* __restrict__ scalars, int32_t* __restrict__ slots)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || i >= scalars[0]) return;
    int c = pt_cell[i];
    if (c < 0) return;
    int r = pt_rank[cell_first[c]];
    if (r >= cfg.max_voxels) return; // unreachable for i < i*, kept as a guard
    int32_t* s = slots + (size_t)r * cfg.max_num_points;
    int x = i;
    for (int k = 0; k < cfg.max_num_points; ++k) {
        int old = atomicMin(&s[k], x);
        x = old > x ? old : x; // the larger of the two moves on
        if (x >= PP_EMPTY) break;
    }
}

__device__ __forceinline__ void vox_gather_body(const float* __restrict__ pts, int nfeat, pp_config cfg,
                                                  const int32_t* __restrict__ slots,
                                                  const int32_t* __restrict__ num_pillars, float* __restrict__ voxels,
                                                  int32_t* __restrict__ npts)
{
    int t = blockIdx.x * blockDim.x + threadIdx.x; // (pillar, slot)
    int T = cfg.max_num_points;
    int p = t / T, k = t - p * T;
    if (p >= *num_pillars) return;
    int idx = slots[t];
    bool have = idx < PP_EMPTY;
    float* dst = voxels + (size_t)t * nfeat;
    if (nfeat == 4) {
        float4 v = have ? *reinterpret_cast<const float4*>(pts + (size_t)idx * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        *reinterpret_cast<float4*>(dst) = v;
    } else {
        for (int f = 0; f < nfeat; ++f) dst[f] = have ? pts[(size_t)idx * nfeat + f] : 0.f;
    }
    if (k == 0) {
        int cnt = 0;
        for (int q = 0; q < T; ++q) cnt += slots[(size_t)p * T + q] < PP_EMPTY;
        npts[p] = cnt;
    }
}

// ---- single-frame entry kernels and their batched twins (blockIdx.z = frame, buffers from the frame table) ----
__global__ void __launch_bounds__(256) vox_cell_first(const float* __restrict__ pts, int n, int nfeat, pp_config cfg,
                                                      int32_t* __restrict__ pt_cell, int32_t* __restrict__ cell_first)
{
    vox_cell_first_body(pts, n, nfeat, cfg, pt_cell, cell_first);
}
__global__ void __launch_bounds__(256) vox_flag_count(const int32_t* __restrict__ pt_cell, const int32_t* __restrict__ cell_first, int n,
                                                      int32_t* __restrict__ wave_cnt)
{
    vox_flag_count_body(pt_cell, cell_first, n, wave_cnt);
}
__global__ void __launch_bounds__(256) vox_rank(const float* __restrict__ pts, int n, int nfeat, pp_config cfg,
                                                const int32_t* __restrict__ pt_cell, const int32_t* __restrict__ cell_first,
                                                const int32_t* __restrict__ wave_cnt, int32_t* __restrict__ pt_rank,
                                                int32_t* __restrict__ coors, int32_t* __restrict__ scalars, int32_t* __restrict__ num_pillars)
{
    vox_rank_body(pts, n, nfeat, cfg, pt_cell, cell_first, wave_cnt, pt_rank, coors, scalars, num_pillars);
}
__global__ void __launch_bounds__(256) vox_insert(int n, pp_config cfg, const int32_t* __restrict__ pt_cell,
                                                  const int32_t* __restrict__ cell_first, const int32_t* __restrict__ pt_rank,
                                                  const int32_t* __restrict__ scalars, int32_t* __restrict__ slots)
{
    vox_insert_body(n, cfg, pt_cell, cell_first, pt_rank, scalars, slots);
}
__global__ void __launch_bounds__(256) vox_gather(const float* __restrict__ pts, int nfeat, pp_config cfg, const int32_t* __restrict__ slots,
                                                  const int32_t* __restrict__ num_pillars, float* __restrict__ voxels, int32_t* __restrict__ npts)
{
    vox_gather_body(pts, nfeat, cfg, slots, num_pillars, voxels, npts);
}

// one fill for everything the integer stages of a frame need initialised: cell_first | slots | scalars (0x7F..),
// the occupancy table (0) and the pillar map (-1)
__global__ void __launch_bounds__(256) pre_init_b(const pp_pre_frame* __restrict__ tab, int n7f, int nocc, int npmap)
{
    const pp_pre_frame F = tab[blockIdx.z];
    const int stride = gridDim.x * blockDim.x;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n7f; i += stride) F.cell_first[i] = PP_EMPTY;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nocc; i += stride) F.occ[i] = 0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < npmap; i += stride) F.pmap[i] = -1;
}
__global__ void __launch_bounds__(256) vox_cell_first_b(pp_in_group in, const pp_pre_frame* __restrict__ tab, int nfeat, pp_config cfg)
{
    const pp_pre_frame F = tab[blockIdx.z];
    vox_cell_first_body(in.pts[blockIdx.z], in.n[blockIdx.z], nfeat, cfg, F.pt_cell, F.cell_first);
}
__global__ void __launch_bounds__(256) vox_flag_count_b(pp_in_group in, const pp_pre_frame* __restrict__ tab)
{
    const pp_pre_frame F = tab[blockIdx.z];
    vox_flag_count_body(F.pt_cell, F.cell_first, in.n[blockIdx.z], F.wave_cnt);
}
__global__ void __launch_bounds__(256) vox_rank_b(pp_in_group in, const pp_pre_frame* __restrict__ tab, int nfeat, pp_config cfg)
{
    const pp_pre_frame F = tab[blockIdx.z];
    vox_rank_body(in.pts[blockIdx.z], in.n[blockIdx.z], nfeat, cfg, F.pt_cell, F.cell_first, F.wave_cnt, F.pt_rank, F.coors, F.scalars, F.num);
}
__global__ void __launch_bounds__(256) vox_insert_b(pp_in_group in, const pp_pre_frame* __restrict__ tab, pp_config cfg)
{
    const pp_pre_frame F = tab[blockIdx.z];
    vox_insert_body(in.n[blockIdx.z], cfg, F.pt_cell, F.cell_first, F.pt_rank, F.scalars, F.slots);
}
__global__ void __launch_bounds__(256) vox_gather_b(pp_in_group in, const pp_pre_frame* __restrict__ tab, int nfeat, pp_config cfg)
{
    const pp_pre_frame F = tab[blockIdx.z];
    vox_gather_body(in.pts[blockIdx.z], nfeat, cfg, F.slots, F.num, F.voxels, F.npts);
}

} // namespace

extern "C" int pp_voxelize(pp_ctx* ctx, const float* pts, int n, int nfeat, float* voxels, int32_t* coors,
                           int32_t* npts, int32_t* num_pillars, void* stream_)
{
    if (!ctx) return PP_E_ARG;
    return pp_voxelize_slot(ctx, 0, pts, n, nfeat, voxels, coors, npts, num_pillars, (hipStream_t)stream_);
}

int pp_voxelize_slot(pp_ctx* ctx, int si, const float* pts, int n, int nfeat, float* voxels, int32_t* coors, int32_t* npts,
                     int32_t* num_pillars, hipStream_t stream)
{
    pp_slot& S = ctx->slot[si];
    const pp_config& cfg = ctx->cfg;
    if (n < 0 || n > cfg.max_points) return pp_fail(ctx, PP_E_ARG, "pp_voxelize: n exceeds cfg.max_points");
    if (nfeat != cfg.num_point_features || nfeat < 3) return pp_fail(ctx, PP_E_ARG, "pp_voxelize: nfeat mismatch");
    if (!voxels || !coors || !npts || !num_pillars || (n > 0 && !pts)) return pp_fail(ctx, PP_E_ARG, "pp_voxelize: null pointer");
    if (n == 0) {
        PP_HIP(hipMemsetAsync(num_pillars, 0, sizeof(int32_t), stream));
        return 0;
    }
    size_t cells = (size_t)cfg.grid_size[0] * cfg.grid_size[1] * cfg.grid_size[2];
    // cell_first | slots | vox_scalars are one allocation (pp_api.hip): a single fill
    PP_HIP(hipMemsetAsync(S.cell_first, 0x7F, (cells + (size_t)cfg.max_voxels * cfg.max_num_points + 4) * sizeof(int32_t), stream));
    int nb = pp_div_up(n, 256);
    hipLaunchKernelGGL(vox_cell_first, dim3(nb), dim3(256), 0, stream, pts, n, nfeat, cfg, S.pt_cell, S.cell_first);
    hipLaunchKernelGGL(vox_flag_count, dim3(nb), dim3(256), 0, stream, S.pt_cell, S.cell_first, n, S.wave_cnt);
    hipLaunchKernelGGL(vox_rank, dim3(nb), dim3(256), 0, stream, pts, n, nfeat, cfg, S.pt_cell, S.cell_first,
                       S.wave_cnt, S.pt_rank, coors, S.vox_scalars, num_pillars);
    hipLaunchKernelGGL(vox_insert, dim3(nb), dim3(256), 0, stream, n, cfg, S.pt_cell, S.cell_first, S.pt_rank,
                       S.vox_scalars, S.slots);
    int nt = cfg.max_voxels * cfg.max_num_points;
    hipLaunchKernelGGL(vox_gather, dim3(pp_div_up(nt, 256)), dim3(256), 0, stream, pts, nfeat, cfg, S.slots, num_pillars,
                       voxels, npts);
    PP_HIP(hipGetLastError());
    return 0;
}

int pp_voxelize_group(pp_ctx* ctx, int b0, int g, const pp_in_group& in, hipStream_t stream)
{
    const pp_config& cfg = ctx->cfg;
    int nmax = 0;
    for (int i = 0; i < g; ++i) {
        if (in.n[i] < 0 || in.n[i] > cfg.max_points) return pp_fail(ctx, PP_E_ARG, "pp_infer_batch: n exceeds cfg.max_points");
        if (in.n[i] > 0 && !in.pts[i]) return pp_fail(ctx, PP_E_ARG, "pp_infer_batch: null point cloud");
        nmax = in.n[i] > nmax ? in.n[i] : nmax;
    }
    const pp_pre_frame* tab = ctx->d_pre + b0;
    const int nfeat = cfg.num_point_features;
    const size_t cells = (size_t)cfg.grid_size[0] * cfg.grid_size[1] * cfg.grid_size[2];
    const int n7f = (int)(cells + (size_t)cfg.max_voxels * cfg.max_num_points + 4);
    const int nbev = ctx->gx * ctx->gy;
    hipLaunchKernelGGL(pre_init_b, dim3(256, 1, g), dim3(256), 0, stream, tab, n7f, nbev, nbev);
    const int nb = pp_div_up(nmax > 0 ? nmax : 1, 256);
    hipLaunchKernelGGL(vox_cell_first_b, dim3(nb, 1, g), dim3(256), 0, stream, in, tab, nfeat, cfg);
    hipLaunchKernelGGL(vox_flag_count_b, dim3(nb, 1, g), dim3(256), 0, stream, in, tab);
    hipLaunchKernelGGL(vox_rank_b, dim3(nb, 1, g), dim3(256), 0, stream, in, tab, nfeat, cfg);
    hipLaunchKernelGGL(vox_insert_b, dim3(nb, 1, g), dim3(256), 0, stream, in, tab, cfg);
    const int nt = cfg.max_voxels * cfg.max_num_points;
    hipLaunchKernelGGL(vox_gather_b, dim3(pp_div_up(nt, 256), 1, g), dim3(256), 0, stream, in, tab, nfeat, cfg);
    PP_HIP(hipGetLastError());
    return 0;
}

// ---- sensor_msgs/PointCloud2 payload -> f32[n,4] (ros_node.py:55-59; sensor_msgs.point_cloud2.read_points) --------
// One thread per (point, field): byte-wise load (fields may sit at any offset), endian swap, convert.  HBM-bound:
// point_step bytes read, 16 bytes written per point.
namespace {
struct pp_pc2_fields { int32_t off[4]; int32_t dt[4]; };

__global__ void __launch_bounds__(256) pc2_unpack(const uint8_t* __restrict__ data, int64_t n, int64_t width, int64_t row_step, int point_step,
                                                  pp_pc2_fields f, int big_endian, float* __restrict__ out)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n * 4) return;
    const int64_t i = t >> 2;
    const int k = (int)(t & 3);
    const uint8_t* src = data + (i / width) * row_step + (i % width) * point_step + f.off[k];
    const int dt = f.dt[k];
    const int nbytes = (dt == 1 || dt == 2) ? 1 : (dt == 3 || dt == 4) ? 2 : (dt == 8) ? 8 : 4;
    uint64_t bits = 0;
    for (int b = 0; b < nbytes; ++b) {
        const uint64_t byte = src[b];
        bits |= byte << (8 * (big_endian ? (nbytes - 1 - b) : b));
    }
    float val;
    switch (dt) {
    case 1: val = (float)(int8_t)bits; break;
    case 2: val = (float)(uint8_t)bits; break;
    case 3: val = (float)(int16_t)bits; break;
    case 4: val = (float)(uint16_t)bits; break;
    case 5: val = (float)(int32_t)bits; break;
    case 6: val = (float)(uint32_t)bits; break;
    case 8: val = (float)__longlong_as_double((long long)bits); break; // numpy .astype(float32): round to nearest
    default: val = __uint_as_float((uint32_t)bits); break;
    }
    out[t] = val;
}
} // namespace

extern "C" int pp_unpack_points(const void* data, int64_t n, int64_t width, int64_t row_step, int point_step, const int32_t* offs,
                                const int32_t* dtypes, int big_endian, float* out, void* stream)
{
    if (n < 0 || width <= 0 || point_step <= 0 || row_step < 0 || !offs || !dtypes) return PP_E_ARG;
    if (n == 0) return 0;
    if (!data || !out) return PP_E_ARG;
    pp_pc2_fields f;
    for (int k = 0; k < 4; ++k) {
        if (dtypes[k] < 1 || dtypes[k] > 8 || offs[k] < 0) return PP_E_ARG;
        const int nb = (dtypes[k] <= 2) ? 1 : (dtypes[k] <= 4) ? 2 : (dtypes[k] == 8) ? 8 : 4;
        if (offs[k] + nb > point_step) return PP_E_ARG;
        f.off[k] = offs[k];
        f.dt[k] = dtypes[k];
    }
    hipLaunchKernelGGL(pc2_unpack, dim3(pp_div_up(n * 4, 256)), dim3(256), 0, (hipStream_t)stream, (const uint8_t*)data, n, width, row_step,
                       point_step, f, big_endian, out);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : -(int)e;
}
