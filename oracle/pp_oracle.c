/* CPU ORACLE (plain C) for the integer / index stages of the PointPillars hot path.
 * TEST INFRASTRUCTURE ONLY: used by tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg; never linked into or called by the product library.
 *
 * Each function restates the reference file:line cited above it.  Pinned by the
 * golden vectors under tests/golden/ (generated from the reference itself by
 * tests/golden/make_goldens.py) and cross-checked against oracle/pp_oracle.py.
 *
 * Build: make -C oracle   (gcc -O2, no -ffast-math: fp32 division/floor must be IEEE)
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* framework/voxel_generator.py:82-106  points_to_voxels
 * scratch: int32[gx*gy*gz] coor_to_voxelidx, filled with -1 by the caller-visible wrapper
 * here (the reference allocates it per frame, :32).  Returns the pillar count. */
int orc_points_to_voxels(const float *pts, int n, int nfeat, const float *voxel_size,
                         const float *offset, const int32_t *grid, int max_voxels,
                         int max_num_points, float *voxels, int32_t *coors, int32_t *num,
                         int32_t *scratch)
{
    const int gx = grid[0], gy = grid[1], gz = grid[2];
    memset(scratch, 0xff, sizeof(int32_t) * (size_t)gx * gy * gz);
    memset(voxels, 0, sizeof(float) * (size_t)max_voxels * max_num_points * nfeat);
    memset(num, 0, sizeof(int32_t) * (size_t)max_voxels);
    memset(coors, 0, sizeof(int32_t) * 3 * (size_t)max_voxels);
    int nv = 0;
    for (int i = 0; i < n; ++i) {
        const float *p = pts + (size_t)i * nfeat;
        int c[3], inside = 1;
        for (int d = 0; d < 3; ++d) {
            float f = floorf((p[d] - offset[d]) / voxel_size[d]); /* true division, :90 */
            if (!(f >= 0.0f && f < (float)grid[d])) { inside = 0; break; }
            c[d] = (int)f;
        }
        if (!inside) continue;
        size_t cell = ((size_t)c[0] * gy + c[1]) * gz + c[2];
        int v = scratch[cell];
        if (v == -1) {
            if (nv >= max_voxels) break; /* :96-97 */
            v = nv++;
            scratch[cell] = v;
            coors[3 * v] = c[0]; coors[3 * v + 1] = c[1]; coors[3 * v + 2] = c[2];
        }
        int k = num[v];
        if (k < max_num_points) {
            memcpy(voxels + ((size_t)v * max_num_points + k) * nfeat, p, sizeof(float) * nfeat);
            num[v] = k + 1;
        }
    }
    return nv;
}

/* framework/anchor_assigner.py:322-335 (CPU path box_np_ops.py:159-165,260-285):
 * occupancy -> cumsum(0).cumsum(1) -> 4-tap lookup without -1 offsets.
 * rects = int32[A,4] (minx,miny,maxx,maxy) from get_anchor_coor (:288-305).
 * scratch: int32[gx*gy]. */
void orc_anchor_mask(const int32_t *coors, int npillars, int gx, int gy, const int32_t *rects,
                     int nanchors, uint8_t *mask, int32_t *scratch)
{
    memset(scratch, 0, sizeof(int32_t) * (size_t)gx * gy);
    for (int p = 0; p < npillars; ++p) scratch[(size_t)coors[3 * p] * gy + coors[3 * p + 1]] += 1;
    for (int x = 1; x < gx; ++x)
        for (int y = 0; y < gy; ++y) scratch[(size_t)x * gy + y] += scratch[(size_t)(x - 1) * gy + y];
    for (int x = 0; x < gx; ++x)
        for (int y = 1; y < gy; ++y) scratch[(size_t)x * gy + y] += scratch[(size_t)x * gy + y - 1];
    for (int a = 0; a < nanchors; ++a) {
        const int32_t *r = rects + 4 * (size_t)a;
        int32_t area = scratch[(size_t)r[2] * gy + r[3]] - scratch[(size_t)r[2] * gy + r[1]]
                     - scratch[(size_t)r[0] * gy + r[3]] + scratch[(size_t)r[0] * gy + r[1]];
        mask[a] = area > 0;
    }
}

/* framework/nms.py:105-116 iou_device ("+1" convention) */
static inline float iou_plus1(const float *a, const float *b)
{
    float left = fmaxf(a[0], b[0]), right = fminf(a[2], b[2]);
    float top = fmaxf(a[1], b[1]), bottom = fminf(a[3], b[3]);
    float w = fmaxf(right - left + 1.0f, 0.0f), h = fmaxf(bottom - top + 1.0f, 0.0f);
    float inter = w * h;
    float sa = (a[2] - a[0] + 1.0f) * (a[3] - a[1] + 1.0f);
    float sb = (b[2] - b[0] + 1.0f) * (b[3] - b[1] + 1.0f);
    return inter / (sa + sb - inter);
}

typedef struct { float s; int32_t i; } orc_key;
static int key_cmp(const void *pa, const void *pb)
{
    const orc_key *a = (const orc_key *)pa, *b = (const orc_key *)pb;
    if (a->s > b->s) return -1;
    if (a->s < b->s) return 1;
    return (a->i > b->i) - (a->i < b->i); /* ties: lower input index first */
}

/* framework/nms.py:6-40 nms_gpu + :119-150 nms_kernel + :85-102 nms_postprocess.
 * dets f32[n,5]; keep gets indices into dets; returns count. */
int orc_nms_aabb(const float *dets, int n, float thresh, int32_t *keep)
{
    if (n <= 0) return 0;
    orc_key *k = (orc_key *)malloc(sizeof(orc_key) * n);
    uint8_t *rem = (uint8_t *)calloc(n, 1);
    for (int i = 0; i < n; ++i) { k[i].s = dets[5 * i + 4]; k[i].i = i; }
    qsort(k, n, sizeof(orc_key), key_cmp);
    int nk = 0;
    for (int i = 0; i < n; ++i) {
        if (rem[i]) continue;
        keep[nk++] = k[i].i;
        const float *a = dets + 5 * (size_t)k[i].i;
        for (int j = i + 1; j < n; ++j)
            if (!rem[j] && iou_plus1(a, dets + 5 * (size_t)k[j].i) > thresh) rem[j] = 1;
    }
    free(k); free(rem);
    return nk;
}

/* ---- rotated IoU, eval/iou.py:164-399 (fp32 scalar code, same operation order) ---- */
static void rb_corners(const float *rb, float *c) /* :351-374 */
{
    float a_cos = cosf(rb[4]), a_sin = sinf(rb[4]);
    float xs[4] = {-rb[2] / 2, -rb[2] / 2, rb[2] / 2, rb[2] / 2};
    float ys[4] = {-rb[3] / 2, rb[3] / 2, rb[3] / 2, -rb[3] / 2};
    for (int i = 0; i < 4; ++i) {
        c[2 * i] = a_cos * xs[i] + a_sin * ys[i] + rb[0];
        c[2 * i + 1] = -a_sin * xs[i] + a_cos * ys[i] + rb[1];
    }
}
static int pt_in_quad(float px, float py, const float *c) /* :308-324 */
{
    float ab0 = c[2] - c[0], ab1 = c[3] - c[1], ad0 = c[6] - c[0], ad1 = c[7] - c[1];
    float ap0 = px - c[0], ap1 = py - c[1];
    float abab = ab0 * ab0 + ab1 * ab1, abap = ab0 * ap0 + ab1 * ap1;
    float adad = ad0 * ad0 + ad1 * ad1, adap = ad0 * ap0 + ad1 * ap1;
    return abab >= abap && abap >= 0 && adad >= adap && adap >= 0;
}
static int seg_inter(const float *p1, const float *p2, int i, int j, float *out) /* :220-263 */
{
    float a0 = p1[2 * i], a1 = p1[2 * i + 1], b0 = p1[2 * ((i + 1) % 4)], b1 = p1[2 * ((i + 1) % 4) + 1];
    float c0 = p2[2 * j], c1 = p2[2 * j + 1], d0 = p2[2 * ((j + 1) % 4)], d1 = p2[2 * ((j + 1) % 4) + 1];
    float ba0 = b0 - a0, ba1 = b1 - a1, da0 = d0 - a0, ca0 = c0 - a0, da1 = d1 - a1, ca1 = c1 - a1;
    int acd = da1 * ca0 > ca1 * da0;
    int bcd = (d1 - b1) * (c0 - b0) > (c1 - b1) * (d0 - b0);
    if (acd != bcd) {
        int abc = ca1 * ba0 > ba1 * ca0, abd = da1 * ba0 > ba1 * da0;
        if (abc != abd) {
            float dc0 = d0 - c0, dc1 = d1 - c1;
            float abba = a0 * b1 - b0 * a1, cddc = c0 * d1 - d0 * c1;
            float dh = ba1 * dc0 - ba0 * dc1;
            out[0] = (abba * dc0 - ba0 * cddc) / dh;
            out[1] = (abba * dc1 - ba1 * cddc) / dh;
            return 1;
        }
    }
    return 0;
}
float orc_rotated_inter(const float *r1, const float *r2) /* inter :377-391 */
{
    float p1[8], p2[8], px[16], py[16], vs[16], t[2];
    int n = 0;
    rb_corners(r1, p1); rb_corners(r2, p2);
    for (int i = 0; i < 4; ++i) { /* :327-348 */
        if (pt_in_quad(p1[2 * i], p1[2 * i + 1], p2)) { px[n] = p1[2 * i]; py[n] = p1[2 * i + 1]; ++n; }
        if (pt_in_quad(p2[2 * i], p2[2 * i + 1], p1)) { px[n] = p2[2 * i]; py[n] = p2[2 * i + 1]; ++n; }
    }
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j)
            if (n < 16 && seg_inter(p1, p2, i, j, t)) { px[n] = t[0]; py[n] = t[1]; ++n; }
    float area = 0.0f;
    if (n > 0) { /* sort_vertex_in_convex_polygon :180-217 */
        float cx = 0, cy = 0;
        for (int i = 0; i < n; ++i) { cx += px[i]; cy += py[i]; }
        cx /= (float)n; cy /= (float)n;
        for (int i = 0; i < n; ++i) {
            float v0 = px[i] - cx, v1 = py[i] - cy, d = sqrtf(v0 * v0 + v1 * v1);
            v0 = v0 / d; v1 = v1 / d;
            if (v1 < 0) v0 = -2 - v0;
            vs[i] = v0;
        }
        for (int i = 1; i < n; ++i)
            if (vs[i - 1] > vs[i]) {
                float tv = vs[i], tx = px[i], ty = py[i];
                int j = i;
                while (j > 0 && vs[j - 1] > tv) { vs[j] = vs[j - 1]; px[j] = px[j - 1]; py[j] = py[j - 1]; --j; }
                vs[j] = tv; px[j] = tx; py[j] = ty;
            }
        for (int i = 0; i < n - 2; ++i) /* area :170-177 */
            area += fabsf(((px[0] - px[i + 2]) * (py[i + 1] - py[i + 2]) - (py[0] - py[i + 2]) * (px[i + 1] - px[i + 2])) / 2.0f);
    }
    return area;
}
float orc_rotated_iou(const float *r1, const float *r2) /* devRotateIoU :394-399 */
{
    float area = orc_rotated_inter(r1, r2);
    return area / (r1[2] * r1[3] + r2[2] * r2[3] - area);
}
/* rotate_iou_gpu_eval, eval/iou.py:540-638: out[i,j] = devRotateIoUEval(query j, box i, criterion) -- the kernel
 * passes the QUERY box first (:600-603), so criterion 0 divides by the query's area, 1 by the box's. */
void orc_rotated_iou_eval(const float *boxes, int n, const float *qboxes, int k, int criterion, float *out)
{
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < k; ++j) {
            const float *r1 = qboxes + 5 * (size_t)j, *r2 = boxes + 5 * (size_t)i;
            float area1 = r1[2] * r1[3], area2 = r2[2] * r2[3];
            float ai = orc_rotated_inter(r1, r2);
            float v;
            if (criterion == -1) v = ai / (area1 + area2 - ai);
            else if (criterion == 0) v = ai / area1;
            else if (criterion == 1) v = ai / area2;
            else v = ai;
            out[(size_t)i * k + j] = v;
        }
}

/* eval/iou.py:438-473 rotate_nms_gpu. dets f32[n,6] (cx,cy,dx,dy,angle,score). */
int orc_nms_rotated(const float *dets, int n, float thresh, int32_t *keep)
{
    if (n <= 0) return 0;
    orc_key *k = (orc_key *)malloc(sizeof(orc_key) * n);
    uint8_t *rem = (uint8_t *)calloc(n, 1);
    for (int i = 0; i < n; ++i) { k[i].s = dets[6 * i + 5]; k[i].i = i; }
    qsort(k, n, sizeof(orc_key), key_cmp);
    int nk = 0;
    for (int i = 0; i < n; ++i) {
        if (rem[i]) continue;
        keep[nk++] = k[i].i;
        for (int j = i + 1; j < n; ++j)
            if (!rem[j] && orc_rotated_iou(dets + 6 * (size_t)k[i].i, dets + 6 * (size_t)k[j].i) > thresh) rem[j] = 1;
    }
    free(k); free(rem);
    return nk;
}
