// Detection post-processing on the device, no host round trips
// (framework/inference.py:26-138 does 4 boolean gathers + topk + 4 D2H + numpy + an NMS round
// trip per class; framework/nms.py:6-150; eval/iou.py:164-473 for the rotated variant).
//
//   Q1 post_filter   per anchor: mask & sigmoid(cls) >= thr -> 64-bit key (score bits | ~anchor)
//                    appended to the class's candidate list + coarse score histogram
//   Q2 post_thresh   per class: histogram suffix scan -> coarse bin holding the K-th best score
//   Q3 post_gather   candidates in bins >= that bin -> short list (K .. K + one bin)
//   Q4 post_topk     per class (one workgroup): exact top-K by bitonic sort of the short list in LDS
//                    (radix-select fallback when a bin is huge), box decode, NMS boxes
//   Q5 nms_mask      64x64 suppression bit tiles, one WAVEFRONT per tile row-block: the 64-bit
//                    ballot of a wave64 IS a mask word (nms.py:119-150 needs a 64-iteration loop)
//   Q6 nms_reduce    greedy sweep, tile-serial (wave-uniform bit tricks inside a tile, vector OR
//                    across tiles), direction flip, range mask, limit_period, compaction
//
// Selection keys make every ordering total: score descending, ties by lower anchor index
// (torch.topk / numpy argsort leave ties unspecified in the reference).
#include <algorithm>
#include <cmath>
#include <cstring>
#include "pp_common.h"

namespace {

constexpr int NBINS = 4096;
constexpr int SHORT_CAP = 4096; // short list capacity per class (K + one coarse bin)
constexpr int MAXK = 4096;      // nms_pre_max upper bound

struct pp_post {
    uint64_t* cand = nullptr;   // [ncls][cand_cap]
    int64_t cand_cap = 0;
    int32_t* counters = nullptr; // [ncls][8]: 0 cand count, 1 short count, 2 thr bin, 3 nsel, 4 nkeep
    int32_t* hist = nullptr;    // [ncls][NBINS]
    uint64_t* shortl = nullptr; // [ncls][SHORT_CAP]
    uint64_t* sel = nullptr;    // [ncls][K] sorted keys
    float* boxes = nullptr;     // [ncls][K][7] decoded
    float* nbox = nullptr;      // [ncls][K][6] NMS boxes (aabb: x1,y1,x2,y2 ; rotated: cx,cy,dx,dy,r)
    int32_t* dirl = nullptr;    // [ncls][K]
    uint64_t* nmask = nullptr;  // [ncls][K][K/64]
    int K = 0, cb = 0, bin_shift = 14;
    uint32_t thr_bits = 0;
};

__device__ __forceinline__ float sigmoid_rn(float x)
{
    // correctly rounded fp32 sigmoid (fp64 inside), same definition as the oracle
    return (float)(1.0 / (1.0 + exp(-(double)x)));
}

// ---------------------------------------------------------------- Q1
constexpr int FILTER_ITEMS = 8; // anchors per thread
__device__ __forceinline__ void post_filter_body(const float* __restrict__ cls, const uint8_t* __restrict__ mask, pp_config cfg,
                                                   float thr, uint32_t thr_bits, int bin_shift, int64_t cand_cap,
                                                   uint64_t* __restrict__ cand, int32_t* __restrict__ counters,
                                                   int32_t* __restrict__ hist)
{
    // * histogram privatised in LDS (scores cluster in a few bins: global atomics on them serialise)
    // * ONE returning global atomic per workgroup for the candidate append (a single counter word
    //   saturates at ~90 returning atomics/us: per-wave appends alone cost > 100 us at 50 % pass rate)
    __shared__ int lh[NBINS];
    __shared__ int wtot[4], wbase[4];
    const int c = blockIdx.y;
    const int begin = cfg.class_begin[c], end = cfg.class_end[c];
    const int base = begin + blockIdx.x * (256 * FILTER_ITEMS);
    if (base >= end) return;
    for (int i = threadIdx.x; i < NBINS; i += 256) lh[i] = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t sb[FILTER_ITEMS];
    int pre[FILTER_ITEMS]; // exclusive position inside the wave, -1 = not a candidate
    int wcount = 0;
#pragma unroll
    for (int it = 0; it < FILTER_ITEMS; ++it) {
        const int a = base + it * 256 + threadIdx.x;
        bool pass = false;
        sb[it] = 0;
        if (a < end && mask[a]) {
            const float s = sigmoid_rn(cls[a]);
            pass = s >= thr;
            sb[it] = __float_as_uint(s);
        }
        const unsigned long long bal = __ballot(pass);
        pre[it] = pass ? wcount + __popcll(bal & ((1ull << lane) - 1ull)) : -1;
        wcount += __popcll(bal);
        if (pass) {
            int bin = (int)((sb[it] - thr_bits) >> bin_shift);
            bin = bin < NBINS ? bin : NBINS - 1;
            atomicAdd(&lh[bin], 1);
        }
    }
    if (lane == 0) wtot[wave] = wcount;
    __syncthreads();
    if (threadIdx.x == 0) {
        const int tot = wtot[0] + wtot[1] + wtot[2] + wtot[3];
        const int b0 = tot ? atomicAdd(&counters[c * 8 + 0], tot) : 0;
        wbase[0] = b0; wbase[1] = b0 + wtot[0]; wbase[2] = wbase[1] + wtot[1]; wbase[3] = wbase[2] + wtot[2];
    }
    __syncthreads();
    const int wb = wbase[wave];
#pragma unroll
    for (int it = 0; it < FILTER_ITEMS; ++it) {
        if (pre[it] < 0) continue;
        const int a = base + it * 256 + threadIdx.x;
        const int slot = wb + pre[it];
        if (slot < cand_cap) cand[(size_t)c * cand_cap + slot] = ((uint64_t)sb[it] << 32) | (uint64_t)(0xFFFFFFFFu - (uint32_t)a);
    }
    for (int i = threadIdx.x; i < NBINS; i += 256) {
        const int v = lh[i];
        if (v) atomicAdd(&hist[c * NBINS + i], v);
    }
}

// ---------------------------------------------------------------- Q2
__device__ __forceinline__ void post_thresh_body(const int32_t* __restrict__ hist, int32_t* __restrict__ counters, int K)
{
    __shared__ int part[1024];
    const int c = blockIdx.x, t = threadIdx.x;
    // thread t owns bins [4t, 4t+4) counted from the TOP
    int v[4], s = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) { v[k] = hist[c * NBINS + (NBINS - 1 - (4 * t + k))]; s += v[k]; }
    part[t] = s;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) { // inclusive Hillis-Steele scan
        int x = (t >= o) ? part[t - o] : 0;
        __syncthreads();
        part[t] += x;
        __syncthreads();
    }
    int before = part[t] - s; // candidates in strictly higher bins than this thread's first
    const int total = part[1023];
    if (total <= K) {
        if (t == 0) counters[c * 8 + 2] = 0; // take everything
        return;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (before < K && before + v[k] >= K) counters[c * 8 + 2] = NBINS - 1 - (4 * t + k);
        before += v[k];
    }
}

// ---------------------------------------------------------------- Q3
constexpr int GATHER_ITEMS = 16; // candidates per thread
__device__ __forceinline__ void post_gather_body(const uint64_t* __restrict__ cand, int64_t cand_cap, int32_t* __restrict__ counters,
                                                   uint32_t thr_bits, int bin_shift, uint64_t* __restrict__ shortl)
{
    // A workgroup scans 4096 candidates, collects the few that reach the threshold bin in LDS and appends them with
    // ONE returning global atomic: per-key appends put thousands of returning atomics on one counter word (and the
    // counters of the frames of a batch share an L2 channel, so they did not even run in parallel across frames).
    __shared__ uint64_t lbuf[1024];
    __shared__ int lcnt, gbase;
    const int c = blockIdx.y, tid = threadIdx.x;
    int n = counters[c * 8 + 0];
    n = n < cand_cap ? n : (int)cand_cap;
    const int base = blockIdx.x * (256 * GATHER_ITEMS);
    if (base >= n) return; // workgroup-uniform
    const int thr_bin = counters[c * 8 + 2];
    if (tid == 0) lcnt = 0;
    __syncthreads();
#pragma unroll 4
    for (int it = 0; it < GATHER_ITEMS; ++it) {
        const int i = base + it * 256 + tid;
        if (i < n) {
            const uint64_t key = cand[(size_t)c * cand_cap + i];
            int bin = (int)(((uint32_t)(key >> 32) - thr_bits) >> bin_shift);
            bin = bin < NBINS ? bin : NBINS - 1;
            if (bin >= thr_bin) {
                const int s = atomicAdd(&lcnt, 1);
                if (s < 1024) {
                    lbuf[s] = key;
                } else { // more than a quarter of the block selected: append directly
                    const int slot = atomicAdd(&counters[c * 8 + 1], 1);
                    if (slot < SHORT_CAP) shortl[(size_t)c * SHORT_CAP + slot] = key;
                }
            }
        }
    }
    __syncthreads();
    const int m = lcnt < 1024 ? lcnt : 1024;
    if (tid == 0 && m > 0) gbase = atomicAdd(&counters[c * 8 + 1], m);
    __syncthreads();
    for (int j = tid; j < m; j += 256) {
        const int slot = gbase + j;
        if (slot < SHORT_CAP) shortl[(size_t)c * SHORT_CAP + slot] = lbuf[j];
    }
}

// in-LDS bitonic sort, DESCENDING, n2 = power of two
__device__ void bitonic_desc(uint64_t* a, int n2)
{
    for (int k = 2; k <= n2; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            __syncthreads();
            for (int i = threadIdx.x; i < n2; i += blockDim.x) {
                int ixj = i ^ j;
                if (ixj > i) {
                    uint64_t x = a[i], y = a[ixj];
                    bool up = ((i & k) == 0); // descending in "up" halves
                    if (up ? (x < y) : (x > y)) { a[i] = y; a[ixj] = x; }
                }
            }
        }
    __syncthreads();
}

__device__ __forceinline__ void decode_box(const float* __restrict__ t, const float* __restrict__ an, float* __restrict__ o)
{
    // box_decode, box_np_ops.py:406-423, fp32, no contraction (file is built with -ffp-contract=off)
    const float xa = an[0], ya = an[1], la = an[3], wa = an[4], ha = an[5], ra = an[6];
    const float za = an[2] + ha / 2.f;
    const float diag = sqrtf(la * la + wa * wa);
    const float xg = t[0] * diag + xa, yg = t[1] * diag + ya, zg = t[2] * ha + za;
    const float lg = (float)exp((double)t[3]) * la, wg = (float)exp((double)t[4]) * wa, hg = (float)exp((double)t[5]) * ha;
    o[0] = xg; o[1] = yg; o[2] = zg - hg / 2.f; o[3] = lg; o[4] = wg; o[5] = hg; o[6] = t[6] + ra;
}

__device__ __forceinline__ void standup_box(float cx, float cy, float dx, float dy, float ang, float* __restrict__ o)
{
    // center_to_corner_box2d + corner_to_standup_nd (box_np_ops.py:64-99,717-726)
    const float s = (float)sin((double)ang), c = (float)cos((double)ang);
    const float sx[4] = {-0.5f, -0.5f, 0.5f, 0.5f}, sy[4] = {-0.5f, 0.5f, 0.5f, -0.5f};
    float x0 = INFINITY, y0 = INFINITY, x1 = -INFINITY, y1 = -INFINITY;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float px = dx * sx[k], py = dy * sy[k];
        const float x = px * c + py * (-s) + cx;
        const float y = px * s + py * c + cy;
        x0 = fminf(x0, x); x1 = fmaxf(x1, x); y0 = fminf(y0, y); y1 = fmaxf(y1, y);
    }
    o[0] = x0; o[1] = y0; o[2] = x1; o[3] = y1;
}

// ---------------------------------------------------------------- Q4
__device__ __forceinline__ void post_topk_body(pp_config cfg, const uint64_t* __restrict__ cand, int64_t cand_cap,
                                                  const uint64_t* __restrict__ shortl, int32_t* __restrict__ counters, int K,
                                                  const float* __restrict__ box, const float* __restrict__ dir,
                                                  const float* __restrict__ anchors, int rotate, uint64_t* __restrict__ sel,
                                                  float* __restrict__ boxes, float* __restrict__ nbox, int32_t* __restrict__ dirl)
{
    __shared__ uint64_t keys[SHORT_CAP];
    __shared__ int s_hist[2048];
    __shared__ uint64_t s_prefix;
    __shared__ int s_need, s_cnt;
    const int c = blockIdx.x;
    int ncand = counters[c * 8 + 0];
    ncand = ncand < cand_cap ? ncand : (int)cand_cap;
    int nshort = counters[c * 8 + 1];
    int n = 0;
    if (nshort <= SHORT_CAP) {
        n = nshort;
        for (int i = threadIdx.x; i < SHORT_CAP; i += blockDim.x) keys[i] = (i < n) ? shortl[(size_t)c * SHORT_CAP + i] : 0ull;
    } else {
        // Fallback (one coarse bin larger than the short list, e.g. thousands of identical scores):
        // exact K-th largest key by MSB-first radix select over the whole candidate list, 11 bits a pass.
        const uint64_t* src = cand + (size_t)c * cand_cap;
        if (threadIdx.x == 0) { s_prefix = 0ull; s_need = K; }
        __syncthreads();
        for (int shift = 53; shift >= -2; shift -= 11) {
            const int sh = shift < 0 ? 0 : shift;
            const int bits = shift < 0 ? 11 + shift : 11;
            const uint64_t himask = (sh + bits >= 64) ? 0ull : (~0ull << (sh + bits));
            for (int i = threadIdx.x; i < 2048; i += blockDim.x) s_hist[i] = 0;
            __syncthreads();
            const uint64_t pf = s_prefix;
            for (int i = threadIdx.x; i < ncand; i += blockDim.x) {
                uint64_t k = src[i];
                if ((k & himask) == (pf & himask)) atomicAdd(&s_hist[(int)((k >> sh) & ((1u << bits) - 1))], 1);
            }
            __syncthreads();
            if (threadIdx.x == 0) {
                int need = s_need, d = (1 << bits) - 1;
                for (; d > 0; --d) {
                    if (s_hist[d] >= need) break;
                    need -= s_hist[d];
                }
                s_need = need;
                s_prefix = pf | ((uint64_t)d << sh);
            }
            __syncthreads();
        }
        const uint64_t kth = s_prefix; // keys are unique, so exactly K keys are >= kth
        if (threadIdx.x == 0) s_cnt = 0;
        __syncthreads();
        for (int i = threadIdx.x; i < ncand; i += blockDim.x) {
            uint64_t k = src[i];
            if (k >= kth) { int sl = atomicAdd(&s_cnt, 1); if (sl < SHORT_CAP) keys[sl] = k; }
        }
        __syncthreads();
        n = s_cnt < SHORT_CAP ? s_cnt : SHORT_CAP;
        for (int i = n + threadIdx.x; i < SHORT_CAP; i += blockDim.x) keys[i] = 0ull;
    }
    __syncthreads();
    int n2 = 64;
    while (n2 < n) n2 <<= 1;
    bitonic_desc(keys, n2);
    const int nsel = n < K ? n : K;
    if (threadIdx.x == 0) counters[c * 8 + 3] = nsel;
    for (int i = threadIdx.x; i < nsel; i += blockDim.x) {
        const uint64_t k = keys[i];
        const uint32_t a = 0xFFFFFFFFu - (uint32_t)(k & 0xFFFFFFFFull);
        sel[(size_t)c * K + i] = k;
        float b[7];
        decode_box(box + (size_t)a * 7, anchors + (size_t)a * 7, b);
        float* bo = boxes + ((size_t)c * K + i) * 7;
#pragma unroll
        for (int q = 0; q < 7; ++q) bo[q] = b[q];
        dirl[(size_t)c * K + i] = dir[(size_t)a * 2 + 1] > dir[(size_t)a * 2] ? 1 : 0;
        float* nb = nbox + ((size_t)c * K + i) * 6;
        if (rotate) { nb[0] = b[0]; nb[1] = b[1]; nb[2] = b[3]; nb[3] = b[4]; nb[4] = b[6]; }
        else standup_box(b[0], b[1], b[3], b[4], b[6], nb);
    }
}

// ---------------------------------------------------------------- IoU device functions
__device__ __forceinline__ float iou_plus1(const float* a, const float* b)
{ // iou_device, nms.py:105-116
    const float left = fmaxf(a[0], b[0]), right = fminf(a[2], b[2]);
    const float top = fmaxf(a[1], b[1]), bottom = fminf(a[3], b[3]);
    const float w = fmaxf(right - left + 1.f, 0.f), h = fmaxf(bottom - top + 1.f, 0.f);
    const float inter = w * h;
    const float sa = (a[2] - a[0] + 1.f) * (a[3] - a[1] + 1.f);
    const float sb = (b[2] - b[0] + 1.f) * (b[3] - b[1] + 1.f);
    return inter / (sa + sb - inter);
}

__device__ __forceinline__ void rb_corners(const float* rb, float* c)
{ // rbbox_to_corners, eval/iou.py:351-374
    const float a_cos = cosf(rb[4]), a_sin = sinf(rb[4]);
    const float xs[4] = {-rb[2] / 2, -rb[2] / 2, rb[2] / 2, rb[2] / 2};
    const float ys[4] = {-rb[3] / 2, rb[3] / 2, rb[3] / 2, -rb[3] / 2};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        c[2 * i] = a_cos * xs[i] + a_sin * ys[i] + rb[0];
        c[2 * i + 1] = -a_sin * xs[i] + a_cos * ys[i] + rb[1];
    }
}
__device__ __forceinline__ bool pt_in_quad(float px, float py, const float* c)
{ // point_in_quadrilateral, eval/iou.py:308-324
    const float ab0 = c[2] - c[0], ab1 = c[3] - c[1], ad0 = c[6] - c[0], ad1 = c[7] - c[1];
    const float ap0 = px - c[0], ap1 = py - c[1];
    const float abab = ab0 * ab0 + ab1 * ab1, abap = ab0 * ap0 + ab1 * ap1;
    const float adad = ad0 * ad0 + ad1 * ad1, adap = ad0 * ap0 + ad1 * ap1;
    return abab >= abap && abap >= 0 && adad >= adap && adap >= 0;
}
__device__ __forceinline__ bool seg_inter(const float* p1, const float* p2, int i, int j, float* out)
{ // line_segment_intersection, eval/iou.py:220-263
    const int i1 = (i + 1) & 3, j1 = (j + 1) & 3;
    const float a0 = p1[2 * i], a1 = p1[2 * i + 1], b0 = p1[2 * i1], b1 = p1[2 * i1 + 1];
    const float c0 = p2[2 * j], c1 = p2[2 * j + 1], d0 = p2[2 * j1], d1 = p2[2 * j1 + 1];
    const float ba0 = b0 - a0, ba1 = b1 - a1, da0 = d0 - a0, ca0 = c0 - a0, da1 = d1 - a1, ca1 = c1 - a1;
    const bool acd = da1 * ca0 > ca1 * da0;
    const bool bcd = (d1 - b1) * (c0 - b0) > (c1 - b1) * (d0 - b0);
    if (acd != bcd) {
        const bool abc = ca1 * ba0 > ba1 * ca0, abd = da1 * ba0 > ba1 * da0;
        if (abc != abd) {
            const float dc0 = d0 - c0, dc1 = d1 - c1;
            const float abba = a0 * b1 - b0 * a1, cddc = c0 * d1 - d0 * c1;
            const float dh = ba1 * dc0 - ba0 * dc1;
            out[0] = (abba * dc0 - ba0 * cddc) / dh;
            out[1] = (abba * dc1 - ba1 * cddc) / dh;
            return true;
        }
    }
    return false;
}
__device__ float rotated_inter_dev(const float* r1, const float* r2)
{ // inter, eval/iou.py:377-391
    float p1[8], p2[8], px[16], py[16], vs[16], t[2];
    int n = 0;
    rb_corners(r1, p1);
    rb_corners(r2, p2);
    for (int i = 0; i < 4; ++i) {
        if (pt_in_quad(p1[2 * i], p1[2 * i + 1], p2)) { px[n] = p1[2 * i]; py[n] = p1[2 * i + 1]; ++n; }
        if (pt_in_quad(p2[2 * i], p2[2 * i + 1], p1)) { px[n] = p2[2 * i]; py[n] = p2[2 * i + 1]; ++n; }
    }
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j)
            if (n < 16 && seg_inter(p1, p2, i, j, t)) { px[n] = t[0]; py[n] = t[1]; ++n; }
    float area = 0.f;
    if (n > 0) { // sort_vertex_in_convex_polygon, :180-217
        float cx = 0.f, cy = 0.f;
        for (int i = 0; i < n; ++i) { cx += px[i]; cy += py[i]; }
        cx /= (float)n; cy /= (float)n;
        for (int i = 0; i < n; ++i) {
            float v0 = px[i] - cx, v1 = py[i] - cy;
            const float d = sqrtf(v0 * v0 + v1 * v1);
            v0 = v0 / d; v1 = v1 / d;
            if (v1 < 0) v0 = -2 - v0;
            vs[i] = v0;
        }
        for (int i = 1; i < n; ++i)
            if (vs[i - 1] > vs[i]) {
                const float tv = vs[i], tx = px[i], ty = py[i];
                int j = i;
                while (j > 0 && vs[j - 1] > tv) { vs[j] = vs[j - 1]; px[j] = px[j - 1]; py[j] = py[j - 1]; --j; }
                vs[j] = tv; px[j] = tx; py[j] = ty;
            }
        for (int i = 0; i < n - 2; ++i) // area, :170-177
            area += fabsf(((px[0] - px[i + 2]) * (py[i + 1] - py[i + 2]) - (py[0] - py[i + 2]) * (px[i + 1] - px[i + 2])) / 2.0f);
    }
    return area;
}
__device__ __forceinline__ float rotated_iou_dev(const float* r1, const float* r2)
{ // devRotateIoU, eval/iou.py:394-399
    const float area = rotated_inter_dev(r1, r2);
    return area / (r1[2] * r1[3] + r2[2] * r2[3] - area);
}

// ---------------------------------------------------------------- Q5
// grid (col tile, row tile, class); 64 threads; lane = COLUMN box, loop over the 64 row boxes;
// the 64-bit ballot of the wave is the mask word of that row.
// TRANSPOSED suppression mask, row-tile major: maskT[rt][col] holds, for column box `col`, the bits r of the row boxes
// (rt*64 + r) that suppress it (IoU > thr and row < col).  One wave per (col tile, row tile); the lane owning a
// column accumulates its own word while the row boxes are broadcast with readlane -- no ballot, no loads in the loop.
__device__ __forceinline__ void nms_mask_body(const int c, const float* __restrict__ nbox, int nstride, const int32_t* __restrict__ nsel_p,
                                               int nsel_stride, int K, int cb, float thr, int rotate, uint64_t* __restrict__ maskT)
{
    const int n = nsel_p[c * nsel_stride];
    const int ct = blockIdx.x, rt = blockIdx.y;
    if (ct < rt || rt * 64 >= n || ct * 64 >= n) return;
    const int lane = threadIdx.x;
    const int col = ct * 64 + lane;
    const float* base = nbox + (size_t)c * K * nstride;
    float cbx[5] = {0, 0, 0, 0, 0}, rmine[5] = {0, 0, 0, 0, 0};
    const int nq = rotate ? 5 : 4;
    if (col < n)
        for (int q = 0; q < nq; ++q) cbx[q] = base[(size_t)col * nstride + q];
    const int myrow = rt * 64 + lane;
    if (myrow < n)
        for (int q = 0; q < nq; ++q) rmine[q] = base[(size_t)myrow * nstride + q];
    const int rows = min(64, n - rt * 64);
    uint64_t word = 0ull;
    for (int r = 0; r < rows; ++r) {
        const int row = rt * 64 + r;
        float rbx[5];
#pragma unroll
        for (int q = 0; q < 5; ++q) rbx[q] = __shfl(rmine[q], r);
        if (col < n && col > row) {
            const float iou = rotate ? rotated_iou_dev(rbx, cbx) : iou_plus1(rbx, cbx);
            if (iou > thr) word |= 1ull << r;
        }
    }
    if (col < n) maskT[((size_t)c * cb + rt) * K + col] = word;
}

// Greedy sweep (nms_postprocess, nms.py:85-102) for one class by ONE wavefront on the transposed mask.
// Lane j keeps the suppression word of tile j.  Per tile t: every lane loads its column's words for tile t
// (diagonal) and the later tiles need only one load + one ballot each.  Returns #kept (<= max_keep).
__device__ __forceinline__ int nms_greedy_wave(const uint64_t* __restrict__ maskT, int n_, int K_, int cb, int max_keep_, int* __restrict__ keep)
{
    // one wave, wave-uniform control: as an out-of-line function its arguments arrived in VGPRs and the whole sweep -- loop control,
    // find-first, the 64-bit masks -- ran as divergent VALU code under exec masking; inlined, with the scalars pinned, it runs on the
    // scalar unit (and the caller's address spaces reach its loads and stores)
    const int n = __builtin_amdgcn_readfirstlane(n_), K = __builtin_amdgcn_readfirstlane(K_), max_keep = __builtin_amdgcn_readfirstlane(max_keep_);
    // maskT[t * K + col]: all loads of a step are 512 contiguous bytes.  The words of row tile t + 1 (diagonal + the next AH column
    // tiles) are requested BEFORE the serial sweep of tile t: nothing in them depends on the sweep, and fetched in place they were a
    // global-memory round trip per tile on the one wave that does the work (16 tiles of a 1000-box class: ~40 of the 59 us).
    constexpr int AH = 16;
    const int lane = threadIdx.x & 63;
    uint64_t remv = 0ull; // lane j: columns of tile j already suppressed
    int nk = 0;
    const int tiles = (n + 63) >> 6;
    if (n <= 0) return 0;
    // branch-free: clamped addresses + selects.  (Written as `cond ? row[i] : 0` every load sat in a basic block of its own and the
    // compiler, unable to count what is in flight, waited for vmcnt(0) in front of every use.)
    auto fetch = [&](int t, uint64_t (&w)[AH], uint64_t& diag) __attribute__((always_inline)) {
        const bool live = t < tiles;
        const uint64_t* row = maskT + (size_t)(live ? t : tiles - 1) * K;
        const int i = t * 64 + lane;
        const uint64_t d = row[i < n ? i : n - 1];
        diag = (live && i < n) ? d : 0ull; // rows of tile t suppressing column i
#pragma unroll
        for (int u = 0; u < AH; ++u) {
            const int cidx = (t + 1 + u) * 64 + lane;
            const uint64_t x = row[cidx < n ? cidx : n - 1];
            w[u] = (live && t + 1 + u < tiles && cidx < n) ? x : 0ull;
        }
    };
    uint64_t wc[AH], diag;
    fetch(0, wc, diag);
    for (int t = 0; t < tiles && nk < max_keep; ++t) {
        uint64_t wn[AH], dn;
        fetch(t + 1, wn, dn);
        const uint64_t* row = maskT + (size_t)t * K;
        const uint64_t rtv = __shfl(remv, t);
        const uint64_t rt = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(rtv >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)rtv);
        const int valid = min(64, n - t * 64);
        uint64_t alive = ~rt & (valid == 64 ? ~0ull : ((1ull << valid) - 1ull));
        uint64_t kept = 0ull;
        while (alive && nk < max_keep) { // wave-uniform
            const int b = __ffsll((long long)alive) - 1;
            kept |= 1ull << b;
            if (lane == 0) keep[nk] = t * 64 + b;
            ++nk;
            alive &= ~(1ull << b);
            alive &= ~__ballot((diag >> b) & 1ull); // columns of this tile suppressed by row b
        }
        if (nk >= max_keep) break;
        // fold: column (j*64+lane) is suppressed if any kept row of tile t suppresses it
#pragma unroll
        for (int u = 0; u < AH; ++u) {
            const unsigned long long bal = __ballot((wc[u] & kept) != 0ull);
            if (lane == t + 1 + u) remv |= bal;
        }
        for (int j0 = t + 1 + AH; j0 < tiles; j0 += 8) { // more than AH tiles ahead (nms_pre_max > 1088): fetched in place
            uint64_t w[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int cidx = (j0 + u) * 64 + lane;
                w[u] = (j0 + u < tiles && cidx < n) ? row[cidx] : 0ull;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const unsigned long long bal = __ballot((w[u] & kept) != 0ull);
                if (lane == j0 + u) remv |= bal;
            }
        }
#pragma unroll
        for (int u = 0; u < AH; ++u) wc[u] = wn[u];
        diag = dn;
    }
    return nk;
}

// ---------------------------------------------------------------- Q6
__device__ __forceinline__ void nms_reduce_body(pp_config cfg, const uint64_t* __restrict__ mask, const uint64_t* __restrict__ sel,
                                                  const float* __restrict__ boxes, const int32_t* __restrict__ dirl,
                                                  int32_t* __restrict__ counters, int K, int cb,
                                                  float* __restrict__ det, int32_t* __restrict__ det_count)
{
    __shared__ int s_cnt[PP_MAX_CLASSES];
    __shared__ int s_out[PP_MAX_CLASSES][MAXK > 1024 ? 1024 : MAXK]; // kept rows surviving the range mask (post_max <= 1024)
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int ncls = cfg.num_classes;
    if (wave < ncls) {
        const int c = wave;
        const int n = __builtin_amdgcn_readfirstlane(counters[c * 8 + 3]);
        // the kept list goes to LDS (compacted in place below): a global store per kept box inside the sweep put an unknown number of
        // stores into the vmcnt queue, so every tile's fold waited for vmcnt(0) -- the NEXT tile's words included (one memory round
        // trip per tile on the one wave that does the work)
        int* keep = s_out[c];
        const int nk = nms_greedy_wave(mask + (size_t)c * K * cb, n, K, cb, cfg.nms_post_max, keep); // <= 1024 (pp_create)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        // direction flip + range mask (quirk: dims vs upper limits, inference.py:107-109), stable compaction
        int no = 0;
        for (int k0 = 0; k0 < nk; k0 += 64) {
            const int k = k0 + lane;
            bool ok = false;
            if (k < nk) {
                const int i = keep[k];
                const float* b = boxes + ((size_t)c * K + i) * 7;
                const bool mn = ((double)b[0] > cfg.center_limit[0]) || ((double)b[1] > cfg.center_limit[1]) || ((double)b[2] > cfg.center_limit[2]);
                const bool mx = ((double)b[3] < cfg.center_limit[3]) || ((double)b[4] < cfg.center_limit[4]) || ((double)b[5] < cfg.center_limit[5]);
                ok = mn && mx;
            }
            const int kv = k < nk ? keep[k] : 0; // every lane reads before any lane writes: positions no + prefix <= k0 + lane
            const unsigned long long bal = __ballot(ok);
            if (ok) s_out[c][no + __popcll(bal & ((1ull << lane) - 1ull))] = kv;
            no += __popcll(bal);
        }
        if (lane == 0) s_cnt[c] = no;
    }
    __syncthreads();
    if (wave < ncls) {
        const int c = wave;
        int off = 0;
        for (int q = 0; q < c; ++q) off += s_cnt[q];
        const int no = s_cnt[c];
        for (int k = lane; k < no; k += 64) {
            const int i = s_out[c][k];
            const float* b = boxes + ((size_t)c * K + i) * 7;
            float r = b[6];
            const bool opp = (r > 0.f) != (dirl[(size_t)c * K + i] != 0);
            if (opp) r = (float)((double)r + 3.141592653589793); // f64 add, f32 store (inference.py:101)
            const float two_pi = 6.283185307179586f;               // limit_period(r, 0.5, 2*pi), fp32
            r = r - floorf(r / two_pi + 0.5f) * two_pi;
            float* d = det + (size_t)(off + k) * 9;
            d[0] = b[0]; d[1] = b[1]; d[2] = b[2]; d[3] = b[3]; d[4] = b[4]; d[5] = b[5]; d[6] = r;
            d[7] = __uint_as_float((uint32_t)(sel[(size_t)c * K + i] >> 32));
            d[8] = (float)c;
        }
        if (lane == 0) det_count[1 + c] = no;
        if (c == ncls - 1 && lane == 0) det_count[0] = off + no;
    }
}

int shift_for(uint32_t thr_bits)
{
    uint32_t range = 0x3F800000u - thr_bits;
    int s = 0;
    while ((range >> s) >= (uint32_t)NBINS) ++s;
    return s;
}

// ---- single-frame entry kernels and their batched twins (blockIdx.z = frame; nms_mask: z = frame * ncls + class) ----
__global__ void __launch_bounds__(256) post_filter(const float* __restrict__ cls, const uint8_t* __restrict__ mask, pp_config cfg, float thr,
                                                   uint32_t thr_bits, int bin_shift, int64_t cand_cap, uint64_t* __restrict__ cand,
                                                   int32_t* __restrict__ counters, int32_t* __restrict__ hist)
{
    post_filter_body(cls, mask, cfg, thr, thr_bits, bin_shift, cand_cap, cand, counters, hist);
}
__global__ void __launch_bounds__(1024) post_thresh(const int32_t* __restrict__ hist, int32_t* __restrict__ counters, int K)
{
    post_thresh_body(hist, counters, K);
}
__global__ void __launch_bounds__(256) post_gather(const uint64_t* __restrict__ cand, int64_t cand_cap, int32_t* __restrict__ counters,
                                                   uint32_t thr_bits, int bin_shift, uint64_t* __restrict__ shortl)
{
    post_gather_body(cand, cand_cap, counters, thr_bits, bin_shift, shortl);
}
__global__ void __launch_bounds__(1024) post_topk(pp_config cfg, const uint64_t* __restrict__ cand, int64_t cand_cap, const uint64_t* __restrict__ shortl,
                                                  int32_t* __restrict__ counters, int K, const float* __restrict__ box, const float* __restrict__ dir,
                                                  const float* __restrict__ anchors, int nms_mode, uint64_t* __restrict__ sel,
                                                  float* __restrict__ boxes, float* __restrict__ nbox, int32_t* __restrict__ dirl)
{
    post_topk_body(cfg, cand, cand_cap, shortl, counters, K, box, dir, anchors, nms_mode, sel, boxes, nbox, dirl);
}
__global__ void __launch_bounds__(64) nms_mask(const float* __restrict__ nbox, int nstride, const int32_t* __restrict__ nsel_p, int nsel_stride,
                                               int K, int cb, float thr, int rotate, uint64_t* __restrict__ maskT)
{
    nms_mask_body(blockIdx.z, nbox, nstride, nsel_p, nsel_stride, K, cb, thr, rotate, maskT);
}
__global__ void __launch_bounds__(64 * PP_MAX_CLASSES) nms_reduce(pp_config cfg, const uint64_t* __restrict__ mask, const uint64_t* __restrict__ sel,
                                                  const float* __restrict__ boxes, const int32_t* __restrict__ dirl, int32_t* __restrict__ counters,
                                                  int K, int cb, float* __restrict__ det, int32_t* __restrict__ det_count)
{
    nms_reduce_body(cfg, mask, sel, boxes, dirl, counters, K, cb, det, det_count);
}

__global__ void __launch_bounds__(256) post_init_b(const pp_post_frame* __restrict__ tab, int nwords)
{
    int32_t* h = tab[blockIdx.z].hist; // hist | counters are one allocation
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nwords; i += gridDim.x * blockDim.x) h[i] = 0;
}
__global__ void __launch_bounds__(256) post_filter_b(const pp_post_frame* __restrict__ tab, pp_config cfg, float thr, uint32_t thr_bits,
                                                     int bin_shift, int64_t cand_cap)
{
    const pp_post_frame F = tab[blockIdx.z];
    post_filter_body(F.cls, F.mask, cfg, thr, thr_bits, bin_shift, cand_cap, F.cand, F.counters, F.hist);
}
__global__ void __launch_bounds__(1024) post_thresh_b(const pp_post_frame* __restrict__ tab, int K)
{
    const pp_post_frame F = tab[blockIdx.z];
    post_thresh_body(F.hist, F.counters, K);
}
__global__ void __launch_bounds__(256) post_gather_b(const pp_post_frame* __restrict__ tab, int64_t cand_cap, uint32_t thr_bits, int bin_shift)
{
    const pp_post_frame F = tab[blockIdx.z];
    post_gather_body(F.cand, cand_cap, F.counters, thr_bits, bin_shift, F.shortl);
}
__global__ void __launch_bounds__(1024) post_topk_b(const pp_post_frame* __restrict__ tab, pp_config cfg, int64_t cand_cap, int K,
                                                    const float* __restrict__ anchors, int nms_mode)
{
    const pp_post_frame F = tab[blockIdx.z];
    post_topk_body(cfg, F.cand, cand_cap, F.shortl, F.counters, K, F.box, F.dir, anchors, nms_mode, F.sel, F.boxes, F.nbox, F.dirl);
}
__global__ void __launch_bounds__(64) nms_mask_b(const pp_post_frame* __restrict__ tab, int ncls, int K, int cb, float thr, int rotate)
{
    const int fr = blockIdx.z / ncls, c = blockIdx.z - fr * ncls;
    const pp_post_frame F = tab[fr];
    nms_mask_body(c, F.nbox, 6, F.counters + 3, 8, K, cb, thr, rotate, F.nmask);
}
__global__ void __launch_bounds__(64 * PP_MAX_CLASSES) nms_reduce_b(const pp_post_frame* __restrict__ tab, pp_config cfg, int K, int cb, float* __restrict__ det,
                                                    size_t det_fs, int32_t* __restrict__ det_count, int cnt_fs)
{
    const pp_post_frame F = tab[blockIdx.z];
    nms_reduce_body(cfg, F.nmask, F.sel, F.boxes, F.dirl, F.counters, K, cb, det + blockIdx.z * det_fs,
                    det_count + blockIdx.z * cnt_fs);
}

// stage entry pp_select_candidates: the sorted keys of post_topk as (anchor id, score) rows, -1 / 0 beyond a class's count
__global__ void __launch_bounds__(256) post_export_sel(const uint64_t* __restrict__ sel, const int32_t* __restrict__ counters, int K,
                                                       int32_t* __restrict__ idx, float* __restrict__ score, int32_t* __restrict__ count)
{
    const int c = blockIdx.x;
    const int n = counters[c * 8 + 3];
    for (int i = threadIdx.x; i < K; i += blockDim.x) {
        const uint64_t k = i < n ? sel[(size_t)c * K + i] : 0ull;
        idx[(size_t)c * K + i] = i < n ? (int32_t)(0xFFFFFFFFu - (uint32_t)(k & 0xFFFFFFFFull)) : -1;
        score[(size_t)c * K + i] = i < n ? __uint_as_float((uint32_t)(k >> 32)) : 0.f;
    }
    if (threadIdx.x == 0) count[c] = n;
}

} // namespace

static int post_create_one(pp_ctx* ctx, pp_slot& S)
{
    pp_post* P = new pp_post();
    S.post = P;
    const pp_config& c = ctx->cfg;
    P->K = c.nms_pre_max;
    P->cb = pp_div_up(P->K, 64);
    int64_t cap = 0;
    for (int i = 0; i < c.num_classes; ++i) {
        if (c.class_end[i] < c.class_begin[i]) return pp_fail(ctx, PP_E_ARG, "bad class range");
        cap = std::max<int64_t>(cap, c.class_end[i] - c.class_begin[i]);
    }
    P->cand_cap = cap > 0 ? cap : 1;
    const int n = c.num_classes;
    PP_HIP(hipMalloc((void**)&P->cand, (size_t)n * P->cand_cap * sizeof(uint64_t)));
    // hist | counters share one allocation: one zero fill per frame
    PP_HIP(hipMalloc((void**)&P->hist, (size_t)n * (NBINS + 8) * sizeof(int32_t)));
    P->counters = P->hist + (size_t)n * NBINS;
    PP_HIP(hipMalloc((void**)&P->shortl, (size_t)n * SHORT_CAP * sizeof(uint64_t)));
    PP_HIP(hipMalloc((void**)&P->sel, (size_t)n * P->K * sizeof(uint64_t)));
    PP_HIP(hipMalloc((void**)&P->boxes, (size_t)n * P->K * 7 * sizeof(float)));
    PP_HIP(hipMalloc((void**)&P->nbox, (size_t)n * P->K * 6 * sizeof(float)));
    PP_HIP(hipMalloc((void**)&P->dirl, (size_t)n * P->K * sizeof(int32_t)));
    PP_HIP(hipMalloc((void**)&P->nmask, (size_t)n * P->K * P->cb * sizeof(uint64_t)));
    float thr = c.score_threshold;
    if (!(thr > 0.f && thr < 1.f)) return pp_fail(ctx, PP_E_ARG, "score_threshold must be in (0,1)");
    memcpy(&P->thr_bits, &thr, 4);
    P->bin_shift = shift_for(P->thr_bits);
    return 0;
}

int pp_post_create(pp_ctx* ctx)
{
    for (pp_slot& S : ctx->slot) {
        int rc = post_create_one(ctx, S);
        if (rc) return rc;
    }
    return 0;
}

void pp_post_destroy(pp_ctx* ctx)
{
    for (pp_slot& S : ctx->slot) {
        pp_post* P = (pp_post*)S.post;
        if (!P) continue;
        void* ptrs[] = {P->cand, P->hist, P->shortl, P->sel, P->boxes, P->nbox, P->dirl, P->nmask};
        for (void* q : ptrs)
            if (q) (void)hipFree(q);
        delete P;
        S.post = nullptr;
    }
}

extern "C" int pp_postprocess(pp_ctx* ctx, const float* cls, const float* box, const float* dir, const uint8_t* mask,
                              float* det, int32_t* det_count, int nms_mode, void* stream_)
{
    if (!ctx) return PP_E_ARG;
    return pp_postprocess_slot(ctx, 0, cls, box, dir, mask, det, det_count, nms_mode, (hipStream_t)stream_);
}

// Stage entry behind Inference.infer_torch (inference.py:140-189: per class mask gather, sigmoid, score threshold, top-k):
// the first two stages of pp_postprocess alone.  idx i32[ncls][nms_pre_max] anchor ids by descending score (ties: lower
// anchor id), score f32[ncls][nms_pre_max], count i32[ncls].
extern "C" int pp_select_candidates(pp_ctx* ctx, const float* cls, const float* box, const float* dir, const uint8_t* mask,
                                    int32_t* idx, float* score, int32_t* count, void* stream_)
{
    if (!ctx) return PP_E_ARG;
    if (!cls || !box || !dir || !mask || !idx || !score || !count) return pp_fail(ctx, PP_E_ARG, "pp_select_candidates: null pointer");
    if (ctx->A == 0) return pp_fail(ctx, PP_E_STATE, "pp_select_candidates: call pp_set_anchors first");
    hipStream_t stream = (hipStream_t)stream_;
    pp_post* P = (pp_post*)ctx->slot[0].post;
    const pp_config& c = ctx->cfg;
    const int n = c.num_classes;
    for (int i = 0; i < n; ++i)
        if (c.class_end[i] > ctx->A) return pp_fail(ctx, PP_E_ARG, "class range exceeds anchor count");
    PP_HIP(hipMemsetAsync(P->hist, 0, (size_t)n * (NBINS + 8) * sizeof(int32_t), stream));
    hipLaunchKernelGGL(post_filter, dim3(pp_div_up(P->cand_cap, 256 * FILTER_ITEMS), n), dim3(256), 0, stream, cls, mask, c, c.score_threshold,
                       P->thr_bits, P->bin_shift, P->cand_cap, P->cand, P->counters, P->hist);
    hipLaunchKernelGGL(post_thresh, dim3(n), dim3(1024), 0, stream, P->hist, P->counters, P->K);
    hipLaunchKernelGGL(post_gather, dim3(pp_div_up(P->cand_cap, 256 * GATHER_ITEMS), n), dim3(256), 0, stream, P->cand, P->cand_cap, P->counters,
                       P->thr_bits, P->bin_shift, P->shortl);
    hipLaunchKernelGGL(post_topk, dim3(n), dim3(1024), 0, stream, c, P->cand, P->cand_cap, P->shortl, P->counters, P->K, box, dir,
                       ctx->anchors, 0, P->sel, P->boxes, P->nbox, P->dirl);
    hipLaunchKernelGGL(post_export_sel, dim3(n), dim3(256), 0, stream, P->sel, P->counters, P->K, idx, score, count);
    PP_HIP(hipGetLastError());
    return 0;
}

int pp_postprocess_slot(pp_ctx* ctx, int si, const float* cls, const float* box, const float* dir, const uint8_t* mask, float* det,
                        int32_t* det_count, int nms_mode, hipStream_t stream)
{
    if (!cls || !box || !dir || !mask || !det || !det_count) return pp_fail(ctx, PP_E_ARG, "pp_postprocess: null pointer");
    if (ctx->A == 0) return pp_fail(ctx, PP_E_STATE, "pp_postprocess: call pp_set_anchors first");
    pp_post* P = (pp_post*)ctx->slot[si].post;
    const pp_config& c = ctx->cfg;
    const int n = c.num_classes;
    for (int i = 0; i < n; ++i)
        if (c.class_end[i] > ctx->A) return pp_fail(ctx, PP_E_ARG, "class range exceeds anchor count");
    { int rc0_ = pp_stage_mark(ctx, stream, PP_ST_POST); if (rc0_) return rc0_; }
    PP_HIP(hipMemsetAsync(P->hist, 0, (size_t)n * (NBINS + 8) * sizeof(int32_t), stream)); // hist + counters
    hipLaunchKernelGGL(post_filter, dim3(pp_div_up(P->cand_cap, 256 * FILTER_ITEMS), n), dim3(256), 0, stream, cls, mask, c, c.score_threshold,
                       P->thr_bits, P->bin_shift, P->cand_cap, P->cand, P->counters, P->hist);
    hipLaunchKernelGGL(post_thresh, dim3(n), dim3(1024), 0, stream, P->hist, P->counters, P->K);
    hipLaunchKernelGGL(post_gather, dim3(pp_div_up(P->cand_cap, 256 * GATHER_ITEMS), n), dim3(256), 0, stream, P->cand, P->cand_cap, P->counters,
                       P->thr_bits, P->bin_shift, P->shortl);
    int rc_;
    if ((rc_ = pp_stage_mark(ctx, stream, PP_ST_POST_TOPK))) return rc_;
    hipLaunchKernelGGL(post_topk, dim3(n), dim3(1024), 0, stream, c, P->cand, P->cand_cap, P->shortl, P->counters, P->K, box, dir,
                       ctx->anchors, nms_mode, P->sel, P->boxes, P->nbox, P->dirl);
    if ((rc_ = pp_stage_mark(ctx, stream, PP_ST_POST_NMS))) return rc_;
    hipLaunchKernelGGL(nms_mask, dim3(P->cb, P->cb, n), dim3(64), 0, stream, P->nbox, 6, P->counters + 3, 8, P->K, P->cb,
                       c.nms_iou_threshold, nms_mode, P->nmask);
    hipLaunchKernelGGL(nms_reduce, dim3(1), dim3(64 * PP_MAX_CLASSES), 0, stream, c, P->nmask, P->sel, P->boxes, P->dirl, P->counters, P->K, P->cb,
                       det, det_count);
    PP_HIP(hipGetLastError());
    return pp_stage_mark(ctx, stream, -1);
}

void pp_post_fill_table(pp_ctx* ctx, int slot, pp_post_frame* f)
{
    const pp_post* P = (const pp_post*)ctx->slot[slot].post;
    PP_SET(f->cand, P->cand); PP_SET(f->shortl, P->shortl); PP_SET(f->sel, P->sel); PP_SET(f->nmask, P->nmask);
    PP_SET(f->counters, P->counters); PP_SET(f->hist, P->hist); PP_SET(f->dirl, P->dirl);
    PP_SET(f->boxes, P->boxes); PP_SET(f->nbox, P->nbox);
}

// post-processing of frames b0 .. b0+g-1 as one launch per stage (blockIdx.z = frame)
int pp_postprocess_group(pp_ctx* ctx, int b0, int g, float* det, int32_t* det_count, int nms_mode, hipStream_t stream)
{
    const pp_post* P = (const pp_post*)ctx->slot[0].post; // sizes are the same for every slot
    const pp_config& c = ctx->cfg;
    const int n = c.num_classes;
    for (int i = 0; i < n; ++i)
        if (c.class_end[i] > ctx->A) return pp_fail(ctx, PP_E_ARG, "class range exceeds anchor count");
    const pp_post_frame* tab = ctx->d_post + b0;
    const size_t det_fs = (size_t)n * c.nms_post_max * 9;
    hipLaunchKernelGGL(post_init_b, dim3(8, 1, g), dim3(256), 0, stream, tab, n * (NBINS + 8));
    hipLaunchKernelGGL(post_filter_b, dim3(pp_div_up(P->cand_cap, 256 * FILTER_ITEMS), n, g), dim3(256), 0, stream, tab, c, c.score_threshold,
                       P->thr_bits, P->bin_shift, P->cand_cap);
    hipLaunchKernelGGL(post_thresh_b, dim3(n, 1, g), dim3(1024), 0, stream, tab, P->K);
    hipLaunchKernelGGL(post_gather_b, dim3(pp_div_up(P->cand_cap, 256 * GATHER_ITEMS), n, g), dim3(256), 0, stream, tab, P->cand_cap, P->thr_bits, P->bin_shift);
    int rc_;
    if ((rc_ = pp_stage_mark(ctx, stream, PP_ST_POST_TOPK))) return rc_;
    hipLaunchKernelGGL(post_topk_b, dim3(n, 1, g), dim3(1024), 0, stream, tab, c, P->cand_cap, P->K, ctx->anchors, nms_mode);
    if ((rc_ = pp_stage_mark(ctx, stream, PP_ST_POST_NMS))) return rc_;
    hipLaunchKernelGGL(nms_mask_b, dim3(P->cb, P->cb, n * g), dim3(64), 0, stream, tab, n, P->K, P->cb, c.nms_iou_threshold, nms_mode);
    hipLaunchKernelGGL(nms_reduce_b, dim3(1, 1, g), dim3(64 * PP_MAX_CLASSES), 0, stream, tab, c, P->K, P->cb, det + (size_t)b0 * det_fs, det_fs,
                       det_count + (size_t)b0 * PP_DET_COUNT_STRIDE, (int)PP_DET_COUNT_STRIDE);
    PP_HIP(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------------
// stateless box ops (framework/box_torch_ops.py:18-77, framework/nms.py:6-40, eval/iou.py:438-473)
// ------------------------------------------------------------------------------------------
namespace {

__global__ void __launch_bounds__(256) k_box_decode(const float* __restrict__ enc, const float* __restrict__ anc, float* __restrict__ out, int64_t n)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float b[7];
    decode_box(enc + i * 7, anc + i * 7, b);
    for (int q = 0; q < 7; ++q) out[i * 7 + q] = b[q];
}

__global__ void __launch_bounds__(256) k_corners2d(const float* __restrict__ ctr, const float* __restrict__ dims,
                                                   const float* __restrict__ ang, float* __restrict__ out, int64_t n)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float dx = dims[2 * i], dy = dims[2 * i + 1], cx = ctr[2 * i], cy = ctr[2 * i + 1];
    const float s = ang ? (float)sin((double)ang[i]) : 0.f, c = ang ? (float)cos((double)ang[i]) : 1.f;
    const float sx[4] = {-0.5f, -0.5f, 0.5f, 0.5f}, sy[4] = {-0.5f, 0.5f, 0.5f, -0.5f};
    for (int k = 0; k < 4; ++k) {
        const float px = dx * sx[k], py = dy * sy[k];
        out[i * 8 + 2 * k] = (ang ? (px * c + py * (-s)) : px) + cx;
        out[i * 8 + 2 * k + 1] = (ang ? (px * s + py * c) : py) + cy;
    }
}

__global__ void __launch_bounds__(256) k_standup2d(const float* __restrict__ cor, float* __restrict__ out, int64_t n)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float* c = cor + i * 8;
    out[i * 4 + 0] = fminf(fminf(c[0], c[2]), fminf(c[4], c[6]));
    out[i * 4 + 1] = fminf(fminf(c[1], c[3]), fminf(c[5], c[7]));
    out[i * 4 + 2] = fmaxf(fmaxf(c[0], c[2]), fmaxf(c[4], c[6]));
    out[i * 4 + 3] = fmaxf(fmaxf(c[1], c[3]), fmaxf(c[5], c[7]));
}

__global__ void __launch_bounds__(256) k_rotated_iou(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ iou, int n, int m)
{
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (int64_t)n * m) return;
    const int i = (int)(t / m), j = (int)(t % m);
    float r1[5], r2[5];
    for (int q = 0; q < 5; ++q) { r1[q] = a[i * 5 + q]; r2[q] = b[j * 5 + q]; }
    iou[t] = rotated_iou_dev(r1, r2);
}

// rotate_iou_kernel_eval, eval/iou.py:563-603: out[i,j] = devRotateIoUEval(query j, box i, criterion)
__global__ void __launch_bounds__(256) k_rotated_iou_eval(const float* __restrict__ boxes, const float* __restrict__ qboxes, float* __restrict__ out,
                                                          int n, int k, int criterion)
{
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (int64_t)n * k) return;
    const int i = (int)(t / k), j = (int)(t % k);
    float r1[5], r2[5];
    for (int q = 0; q < 5; ++q) { r1[q] = qboxes[j * 5 + q]; r2[q] = boxes[i * 5 + q]; }
    const float area1 = r1[2] * r1[3], area2 = r2[2] * r2[3];
    const float ai = rotated_inter_dev(r1, r2);
    float v;
    if (criterion == -1) v = ai / (area1 + area2 - ai);
    else if (criterion == 0) v = ai / area1;
    else if (criterion == 1) v = ai / area2;
    else v = ai;
    out[t] = v;
}

// sort dets by score (desc, ties by lower index) into nbox[n][6] (+ order[n]); one workgroup, n <= MAXK
__global__ void __launch_bounds__(1024) k_nms_sort(const float* __restrict__ dets, int n, int stride, float* __restrict__ nbox,
                                                   int32_t* __restrict__ order, int32_t* __restrict__ nsel)
{
    __shared__ uint64_t keys[MAXK];
    int n2 = 64;
    while (n2 < n) n2 <<= 1;
    for (int i = threadIdx.x; i < n2; i += blockDim.x) {
        uint64_t k = 0ull;
        if (i < n) {
            uint32_t u = __float_as_uint(dets[(size_t)i * stride + stride - 1]);
            u = (u & 0x80000000u) ? ~u : (u | 0x80000000u); // total order on floats
            k = ((uint64_t)u << 32) | (uint64_t)(0xFFFFFFFFu - (uint32_t)i);
            if (k == 0ull) k = 1ull;
        }
        keys[i] = k;
    }
    bitonic_desc(keys, n2);
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const uint32_t src = 0xFFFFFFFFu - (uint32_t)(keys[i] & 0xFFFFFFFFull);
        order[i] = (int32_t)src;
        for (int q = 0; q < stride - 1; ++q) nbox[(size_t)i * 6 + q] = dets[(size_t)src * stride + q];
    }
    if (threadIdx.x == 0) *nsel = n;
}

__global__ void __launch_bounds__(64) k_nms_reduce(const uint64_t* __restrict__ mask, const int32_t* __restrict__ order, int n, int cb,
                                                   int32_t* __restrict__ tmp, int32_t* __restrict__ keep, int32_t* __restrict__ nkeep)
{
    const int nk = nms_greedy_wave(mask, n, n, cb, n, tmp);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    for (int k = threadIdx.x; k < nk; k += 64) keep[k] = order[tmp[k]];
    if (threadIdx.x == 0) *nkeep = nk;
}

} // namespace

extern "C" int pp_box_decode(const float* enc, const float* anchors, float* out, int64_t n, void* stream)
{
    if (n < 0 || (n > 0 && (!enc || !anchors || !out))) return PP_E_ARG;
    if (n == 0) return 0;
    hipLaunchKernelGGL(k_box_decode, dim3(pp_div_up(n, 256)), dim3(256), 0, (hipStream_t)stream, enc, anchors, out, n);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : -(int)e;
}

extern "C" int pp_corners2d(const float* centers, const float* dims, const float* angles, float* corners, int64_t n, void* stream)
{
    if (n < 0 || (n > 0 && (!centers || !dims || !corners))) return PP_E_ARG;
    if (n == 0) return 0;
    hipLaunchKernelGGL(k_corners2d, dim3(pp_div_up(n, 256)), dim3(256), 0, (hipStream_t)stream, centers, dims, angles, corners, n);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : -(int)e;
}

extern "C" int pp_standup2d(const float* corners, float* boxes, int64_t n, void* stream)
{
    if (n < 0 || (n > 0 && (!corners || !boxes))) return PP_E_ARG;
    if (n == 0) return 0;
    hipLaunchKernelGGL(k_standup2d, dim3(pp_div_up(n, 256)), dim3(256), 0, (hipStream_t)stream, corners, boxes, n);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : -(int)e;
}

extern "C" int pp_rotated_iou(const float* a, const float* b, float* iou, int n, int m, void* stream)
{
    if (n < 0 || m < 0 || ((int64_t)n * m > 0 && (!a || !b || !iou))) return PP_E_ARG;
    if ((int64_t)n * m == 0) return 0;
    hipLaunchKernelGGL(k_rotated_iou, dim3(pp_div_up((int64_t)n * m, 256)), dim3(256), 0, (hipStream_t)stream, a, b, iou, n, m);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : -(int)e;
}

extern "C" int pp_rotated_iou_eval(const float* boxes, const float* qboxes, float* out, int n, int k, int criterion, void* stream)
{
    if (n < 0 || k < 0 || criterion < -1 || criterion > 2) return PP_E_ARG;
    if (n == 0 || k == 0) return 0;
    if (!boxes || !qboxes || !out) return PP_E_ARG;
    hipLaunchKernelGGL(k_rotated_iou_eval, dim3(pp_div_up((int64_t)n * k, 256)), dim3(256), 0, (hipStream_t)stream, boxes, qboxes, out, n, k, criterion);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : -(int)e;
}

extern "C" int pp_nms(const float* dets, int n, int stride, float thresh, int32_t* keep, int32_t* nkeep, int rotate, void* stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    if (n < 0 || n > MAXK || !nkeep || (n > 0 && (!dets || !keep))) return PP_E_ARG;
    if ((rotate && stride != 6) || (!rotate && stride != 5)) return PP_E_ARG;
    if (n == 0) {
        hipError_t e0 = hipMemsetAsync(nkeep, 0, sizeof(int32_t), stream);
        return e0 == hipSuccess ? 0 : -(int)e0;
    }
    const int cb = pp_div_up(n, 64);
    // stream-ordered scratch: sorted boxes, order, tmp keep, nsel, mask
    size_t bytes = (size_t)n * 6 * 4 + (size_t)n * 4 * 2 + 64 + (size_t)n * cb * 8 + 256;
    char* ws = nullptr;
    hipError_t e = hipMallocAsync((void**)&ws, bytes, stream);
    if (e != hipSuccess) return -(int)e;
    uint64_t* mask = (uint64_t*)ws;
    float* nbox = (float*)(ws + (size_t)n * cb * 8);
    int32_t* order = (int32_t*)(nbox + (size_t)n * 6);
    int32_t* tmp = order + n;
    int32_t* nsel = tmp + n;
    hipLaunchKernelGGL(k_nms_sort, dim3(1), dim3(1024), 0, stream, dets, n, stride, nbox, order, nsel);
    hipLaunchKernelGGL(nms_mask, dim3(cb, cb, 1), dim3(64), 0, stream, nbox, 6, nsel, 0, n, cb, thresh, rotate, mask);
    hipLaunchKernelGGL(k_nms_reduce, dim3(1), dim3(64), 0, stream, mask, order, n, cb, tmp, keep, nkeep);
    e = hipGetLastError();
    hipError_t e2 = hipFreeAsync(ws, stream);
    if (e != hipSuccess) return -(int)e;
    return e2 == hipSuccess ? 0 : -(int)e2;
}
