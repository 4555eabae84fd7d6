"""CPU ORACLE for the PointPillars inference hot path -- TEST INFRASTRUCTURE ONLY.

This file restates, on the CPU, the algorithm of the reference
(1005088h/3d_object_detection) for the path
voxelise -> anchor mask -> PFN -> BEV scatter -> backbone -> head -> decode -> NMS.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
it; the product (3d_object_detection_amd/) never does.

Pinning: the reference ships no tests or golden vectors.  The
oracle is pinned against outputs of the reference itself, run in the build
container and committed under tests/golden/ by tests/golden/make_goldens.py
(integer stages bit-exact, float stages <= 1e-5).  The numba.cuda kernel bodies
(nms_kernel, rotate_nms_kernel, get_anchors_mask_gpu) cannot execute anywhere
here; for those the oracle follows the source text cited at each function and the
goldens come from the reference's equivalent CPU path / device functions run as
plain Python.

Integer / index work is numpy; floating-point network stages use torch CPU fp32
functional ops (the same ATen kernels the reference's nn.Modules dispatch to).
"""
import math

import numpy as np

F32 = np.float32


# ----------------------------------------------------------------------------
# a1  VoxelGenerator.__init__            framework/voxel_generator.py:6-26
# ----------------------------------------------------------------------------
def voxel_setup(config):
    """Snap the detection range to whole cells.  Returns dict with voxel_size f32[3],
    offset f32[3], grid_size i32[3], detection_range f32[6], range_diff f32[3]."""
    dr = np.asarray(config["detection_range"], dtype=F32)
    lo, hi = dr[:3], dr[3:]
    center = (hi + lo) / 2
    vs = np.asarray(config["voxel_size"], dtype=F32)
    grid = ((hi - lo) / vs).astype(np.int32)
    range_diff = grid.astype(F32) * vs
    offset = center - range_diff / 2
    return {
        "voxel_size": vs,
        "offset": offset.astype(F32),
        "grid_size": grid,
        "detection_range": np.concatenate([offset, offset + range_diff]).astype(F32),
        "range_diff": range_diff.astype(F32),
    }


# ----------------------------------------------------------------------------
# a2  points_to_voxels                   framework/voxel_generator.py:82-106
# ----------------------------------------------------------------------------
def points_to_voxels(points, voxel_size, offset, grid_size, max_voxels, max_num_points):
    """Sequential hash-and-gather.  Pillar id = order of first appearance; each
    pillar keeps its first max_num_points points in stream order; the loop BREAKS
    at the (max_voxels+1)-th new pillar (:96-97) -- every later point is dropped
    even when its pillar already exists.
    Returns voxels f32[P,T,F] (zero padded), coors i32[P,3] (x,y,z), num i32[P]."""
    points = np.ascontiguousarray(points, dtype=F32)
    n, f = points.shape
    gx, gy, gz = (int(v) for v in grid_size)
    # IEEE fp32 true division then floor (:90); no reciprocal multiply.
    cf = np.floor((points[:, :3] - offset.astype(F32)) / voxel_size.astype(F32))
    inside = ((cf >= 0) & (cf < np.asarray([gx, gy, gz], dtype=F32))).all(axis=1)
    # non-finite coordinates are treated as outside (the reference's int cast is UB there)
    inside &= np.isfinite(cf).all(axis=1)
    ci = np.where(inside[:, None], cf, 0).astype(np.int64)
    cell = (ci[:, 0] * gy + ci[:, 1]) * gz + ci[:, 2]
    voxels = np.zeros((max_voxels, max_num_points, f), dtype=F32)
    num = np.zeros((max_voxels,), dtype=np.int32)
    coors = np.zeros((max_voxels, 3), dtype=np.int32)
    lut = {}
    nv = 0
    for i in range(n):
        if not inside[i]:
            continue
        c = int(cell[i])
        v = lut.get(c, -1)
        if v < 0:
            if nv >= max_voxels:
                break
            v = nv
            lut[c] = v
            coors[v] = ci[i]
            nv += 1
        k = num[v]
        if k < max_num_points:
            voxels[v, k] = points[i]
            num[v] = k + 1
    return voxels[:nv], coors[:nv], num[:nv]


# ----------------------------------------------------------------------------
# a3  AnchorAssigner.__init__/.generate  framework/anchor_assigner.py:221-320
#     rbbox2d_to_near_bbox / get_anchor_coor   framework/box_np_ops.py:308-320,288-305
# ----------------------------------------------------------------------------
# class table hard-coded by the reference at anchor_assigner.py:222-245
ANCHOR_CLASSES = (
    ("vehicle", ((4.6, 2.10, 1.8), (7.5, 2.6, 2.9), (12.6, 2.9, 3.8)), (0, 1.5707963267948966)),
    ("pedestrian", ((0.96874749, 0.9645992, 1.81212425),), (0,)),
    ("cyclist", ((2.02032733, 0.98075615, 1.72027404),), (0, 1.5707963267948966)),
)


def limit_period(val, offset=0.5, period=np.pi):
    """box_np_ops.py:102-103."""
    return val - np.floor(val / period + offset) * period


def _one_anchor_grid(size, rot, fmap, strides, offsets):
    """anchor_assigner.py:300-320: centres = offset + stride/2 + i*stride, z = h/2."""
    fx, fy = int(fmap[0]), int(fmap[1])
    xs = np.arange(fx, dtype=F32) * strides[0] + (offsets[0] + strides[0] / 2)
    ys = np.arange(fy, dtype=F32) * strides[1] + (offsets[1] + strides[1] / 2)
    zc = F32(np.arange(1, dtype=F32)[0] * strides[2] + size[2] / 2)
    out = np.empty((fx, fy, 7), dtype=F32)
    out[:, :, 0] = xs[:, None]
    out[:, :, 1] = ys[None, :]
    out[:, :, 2] = zc
    out[:, :, 3:6] = np.asarray(size, dtype=F32)
    out[:, :, 6] = F32(rot)
    return out.reshape(-1, 7)


def near_bbox(rb):
    """box_np_ops.py:308-320 rbbox2d_to_near_bbox on [N,5] (x,y,dx,dy,r) -> [N,4]."""
    rots = rb[:, 4]
    swap = np.abs(limit_period(rots, 0.5, np.pi)) > np.pi / 4
    dx = np.where(swap, rb[:, 3], rb[:, 2])
    dy = np.where(swap, rb[:, 2], rb[:, 3])
    return np.stack([rb[:, 0] - dx / 2, rb[:, 1] - dy / 2, rb[:, 0] + dx / 2, rb[:, 1] + dy / 2], axis=1).astype(F32)


def anchor_cell_rects(anchors_bv, voxel_size, offset, grid_size):
    """box_np_ops.py:288-305 get_anchor_coor: clamped cell rectangle per anchor."""
    vs = voxel_size.astype(F32)
    off = offset.astype(F32)
    c = np.empty(anchors_bv.shape, dtype=np.int32)
    c[:, 0] = np.maximum(np.floor((anchors_bv[:, 0] - off[0]) / vs[0]).astype(np.int32), 0)
    c[:, 1] = np.maximum(np.floor((anchors_bv[:, 1] - off[1]) / vs[1]).astype(np.int32), 0)
    c[:, 2] = np.minimum(np.floor((anchors_bv[:, 2] - off[0]) / vs[0]).astype(np.int32), int(grid_size[0]) - 1)
    c[:, 3] = np.minimum(np.floor((anchors_bv[:, 3] - off[1]) / vs[1]).astype(np.int32), int(grid_size[1]) - 1)
    return c


def make_anchors(setup, feature_map_size=None, class_table=None):
    """Anchor table in the reference's order class -> size -> rotation -> x -> y.
    The reference hard-codes a 400x400 map (anchor_assigner.py:227); the oracle
    derives it as grid/2 (identical for eight_20cm) so the other configs work.
    class_table (dict name -> {sizes, rotations}, insertion-ordered) replaces the reference's hard-coded three
    classes for the build-side 10-class nuScenes config, which has no reference counterpart (parity unpinned there)."""
    classes = ANCHOR_CLASSES if not class_table else [(n, t["sizes"], t["rotations"]) for n, t in class_table.items()]
    grid = setup["grid_size"]
    if feature_map_size is None:
        feature_map_size = [int(grid[0]) // 2, int(grid[1]) // 2, 1]
    fmap = np.asarray(feature_map_size, dtype=F32)
    strides = setup["range_diff"] / fmap
    tabs, masks, start = [], {}, 0
    for name, sizes, rots in classes:
        parts = [_one_anchor_grid(s, r, fmap, strides, setup["offset"]) for s in sizes for r in rots]
        t = np.concatenate(parts)
        tabs.append(t)
        masks[name] = [start, start + t.shape[0]]
        start += t.shape[0]
    anchors = np.concatenate(tabs)
    bv = near_bbox(anchors[:, [0, 1, 3, 4, 6]])
    rects = anchor_cell_rects(bv, setup["voxel_size"], setup["offset"], grid)
    return {"anchors": anchors, "class_masks": masks, "anchors_bv": bv, "anchors_coors": rects,
            "feature_map_size": [int(v) for v in feature_map_size]}


# ----------------------------------------------------------------------------
# a4  AnchorAssigner.create_mask         framework/anchor_assigner.py:322-335
#     (CPU path box_np_ops.py:159-165,260-285 == CUDA path :168-257)
# ----------------------------------------------------------------------------
def create_mask(coors, grid_size, anchors_coors):
    """Occupancy integral image; anchor kept if its clamped cell-rect 4-tap sum > 0.
    Note the reference's summed-area lookup has no -1 offsets (box_np_ops.py:278-283)."""
    gx, gy = int(grid_size[0]), int(grid_size[1])
    dense = np.zeros((gx, gy), dtype=np.int64)
    np.add.at(dense, (coors[:, 0], coors[:, 1]), 1)
    dense = dense.cumsum(0).cumsum(1)
    r = anchors_coors
    area = dense[r[:, 2], r[:, 3]] - dense[r[:, 2], r[:, 1]] - dense[r[:, 0], r[:, 3]] + dense[r[:, 0], r[:, 1]]
    return area > 0


# ----------------------------------------------------------------------------
# a11/a12 box math                       framework/box_np_ops.py:406-423,64-99,122-153,717-726
# ----------------------------------------------------------------------------
def box_decode(enc, anchors):
    """(x,y,z,l,w,h,r) residual decode, box_np_ops.py:406-423 == box_torch_ops.py:61-77."""
    enc = np.asarray(enc, dtype=F32)
    anchors = np.asarray(anchors, dtype=F32)
    xa, ya, za, la, wa, ha, ra = (anchors[..., i] for i in range(7))
    xt, yt, zt, lt, wt, ht, rt = (enc[..., i] for i in range(7))
    za = za + ha / 2
    diag = np.sqrt(la ** 2 + wa ** 2)
    xg = xt * diag + xa
    yg = yt * diag + ya
    zg = zt * ha + za
    lg = np.exp(lt) * la
    wg = np.exp(wt) * wa
    hg = np.exp(ht) * ha
    rg = rt + ra
    zg = zg - hg / 2
    return np.stack([xg, yg, zg, lg, wg, hg, rg], axis=-1).astype(F32)


_CORNER_SIGNS = np.asarray([[-0.5, -0.5], [-0.5, 0.5], [0.5, 0.5], [0.5, -0.5]], dtype=F32)


def center_to_corner_box2d(centers, dims, angles=None):
    """Corners (clockwise from the minimum corner) rotated clockwise-positive
    (box_np_ops.py:64-99,122-153; origin 0.5). Returns f32[N,4,2]."""
    centers = np.asarray(centers, dtype=F32)
    dims = np.asarray(dims, dtype=F32)
    corners = dims[:, None, :] * _CORNER_SIGNS[None, :, :]
    if angles is not None:
        s = np.sin(angles).astype(F32)
        c = np.cos(angles).astype(F32)
        # rot_mat_T = [[c, s], [-s, c]]; out[a,i,k] = sum_j p[a,i,j] * R[j,k,a]
        x = corners[:, :, 0] * c[:, None] + corners[:, :, 1] * (-s)[:, None]
        y = corners[:, :, 0] * s[:, None] + corners[:, :, 1] * c[:, None]
        corners = np.stack([x, y], axis=-1).astype(F32)
    return (corners + centers[:, None, :]).astype(F32)


def corner_to_standup_nd(corners):
    """box_np_ops.py:717-726: (min x, min y, max x, max y)."""
    return np.concatenate([corners.min(axis=1), corners.max(axis=1)], axis=-1).astype(F32)


# ----------------------------------------------------------------------------
# a13  nms_gpu / nms_kernel / iou_device / nms_postprocess   framework/nms.py:6-150
# ----------------------------------------------------------------------------
def _order_desc(scores):
    """Sort by score descending; ties by lower input index first (the reference's
    argsort()[::-1] leaves tie order unspecified -- we fix it, documented in DESIGN.md)."""
    idx = np.arange(scores.shape[0])
    return np.lexsort((idx, -scores.astype(np.float64))).astype(np.int32)


def aabb_iou_plus1(a, b):
    """iou_device, nms.py:105-116: legacy '+1' pixel convention, fp32."""
    one = F32(1.0)
    w = np.maximum(np.minimum(a[2], b[2]) - np.maximum(a[0], b[0]) + one, F32(0))
    h = np.maximum(np.minimum(a[3], b[3]) - np.maximum(a[1], b[1]) + one, F32(0))
    inter = w * h
    sa = (a[2] - a[0] + one) * (a[3] - a[1] + one)
    sb = (b[2] - b[0] + one) * (b[3] - b[1] + one)
    return inter / (sa + sb - inter)


def _greedy(sup):
    """nms_postprocess, nms.py:85-102, on a dense suppression matrix sup[i,j] (j>i)."""
    n = sup.shape[0]
    removed = np.zeros(n, dtype=bool)
    keep = []
    for i in range(n):
        if not removed[i]:
            keep.append(i)
            removed |= sup[i]
    return keep


def nms_aabb(dets, thresh):
    """nms_gpu, nms.py:6-40. dets f32[n,5] (x1,y1,x2,y2,score) -> indices into dets."""
    dets = np.asarray(dets, dtype=F32)
    n = dets.shape[0]
    if n == 0:
        return []
    order = _order_desc(dets[:, 4])
    b = dets[order]
    one = F32(1.0)
    x1, y1, x2, y2 = b[:, 0], b[:, 1], b[:, 2], b[:, 3]
    w = np.maximum(np.minimum(x2[:, None], x2[None, :]) - np.maximum(x1[:, None], x1[None, :]) + one, F32(0))
    h = np.maximum(np.minimum(y2[:, None], y2[None, :]) - np.maximum(y1[:, None], y1[None, :]) + one, F32(0))
    inter = (w * h).astype(F32)
    area = ((x2 - x1 + one) * (y2 - y1 + one)).astype(F32)
    iou = inter / (area[:, None] + area[None, :] - inter)
    sup = np.triu(iou > F32(thresh), k=1)
    return [int(order[i]) for i in _greedy(sup)]


# ----------------------------------------------------------------------------
# a14  rotate_nms_gpu + device functions  eval/iou.py:164-473   (fp32 scalar code)
# ----------------------------------------------------------------------------
def _rbbox_corners(rb):
    """rbbox_to_corners, eval/iou.py:351-374."""
    a_cos = F32(math.cos(rb[4]))
    a_sin = F32(math.sin(rb[4]))
    cx, cy, xd, yd = rb[0], rb[1], rb[2], rb[3]
    xs = (-xd / F32(2), -xd / F32(2), xd / F32(2), xd / F32(2))
    ys = (-yd / F32(2), yd / F32(2), yd / F32(2), -yd / F32(2))
    out = np.empty(8, dtype=F32)
    for i in range(4):
        out[2 * i] = a_cos * xs[i] + a_sin * ys[i] + cx
        out[2 * i + 1] = -a_sin * xs[i] + a_cos * ys[i] + cy
    return out


def _pt_in_quad(px, py, c):
    """point_in_quadrilateral, eval/iou.py:308-324."""
    ab0, ab1 = c[2] - c[0], c[3] - c[1]
    ad0, ad1 = c[6] - c[0], c[7] - c[1]
    ap0, ap1 = px - c[0], py - c[1]
    abab = ab0 * ab0 + ab1 * ab1
    abap = ab0 * ap0 + ab1 * ap1
    adad = ad0 * ad0 + ad1 * ad1
    adap = ad0 * ap0 + ad1 * ap1
    return abab >= abap and abap >= 0 and adad >= adap and adap >= 0


def _seg_inter(p1, p2, i, j):
    """line_segment_intersection, eval/iou.py:220-263. Returns (x,y) or None."""
    a0, a1 = p1[2 * i], p1[2 * i + 1]
    b0, b1 = p1[2 * ((i + 1) % 4)], p1[2 * ((i + 1) % 4) + 1]
    c0, c1 = p2[2 * j], p2[2 * j + 1]
    d0, d1 = p2[2 * ((j + 1) % 4)], p2[2 * ((j + 1) % 4) + 1]
    ba0, ba1 = b0 - a0, b1 - a1
    da0, ca0 = d0 - a0, c0 - a0
    da1, ca1 = d1 - a1, c1 - a1
    acd = da1 * ca0 > ca1 * da0
    bcd = (d1 - b1) * (c0 - b0) > (c1 - b1) * (d0 - b0)
    if acd != bcd:
        abc = ca1 * ba0 > ba1 * ca0
        abd = da1 * ba0 > ba1 * da0
        if abc != abd:
            dc0, dc1 = d0 - c0, d1 - c1
            abba = a0 * b1 - b0 * a1
            cddc = c0 * d1 - d0 * c1
            dh = ba1 * dc0 - ba0 * dc1
            dx = abba * dc0 - ba0 * cddc
            dy = abba * dc1 - ba1 * cddc
            return dx / dh, dy / dh
    return None


def rotated_inter_area(rb1, rb2):
    """inter(), eval/iou.py:377-391: clip two rotated rects, angular sort, fan area."""
    with np.errstate(all="ignore"):
        p1 = _rbbox_corners(rb1)
        p2 = _rbbox_corners(rb2)
        pts = []
        for i in range(4):  # quadrilateral_intersection :327-348
            if _pt_in_quad(p1[2 * i], p1[2 * i + 1], p2):
                pts.append((p1[2 * i], p1[2 * i + 1]))
            if _pt_in_quad(p2[2 * i], p2[2 * i + 1], p1):
                pts.append((p2[2 * i], p2[2 * i + 1]))
        for i in range(4):
            for j in range(4):
                r = _seg_inter(p1, p2, i, j)
                if r is not None:
                    pts.append(r)
        n = len(pts)
        if n == 0:
            return F32(0)
        px = np.asarray([p[0] for p in pts], dtype=F32)
        py = np.asarray([p[1] for p in pts], dtype=F32)
        # sort_vertex_in_convex_polygon :180-217 (insertion sort on a pseudo-angle)
        cx = F32(0)
        cy = F32(0)
        for k in range(n):
            cx = F32(cx + px[k])
            cy = F32(cy + py[k])
        cx = F32(cx / F32(n))
        cy = F32(cy / F32(n))
        vs = np.empty(n, dtype=F32)
        for k in range(n):
            v0 = F32(px[k] - cx)
            v1 = F32(py[k] - cy)
            d = F32(math.sqrt(F32(v0 * v0 + v1 * v1)))
            v0 = F32(v0 / d)
            v1 = F32(v1 / d)
            if v1 < 0:
                v0 = F32(F32(-2) - v0)
            vs[k] = v0
        for k in range(1, n):
            if vs[k - 1] > vs[k]:
                t, tx, ty = vs[k], px[k], py[k]
                j = k
                while j > 0 and vs[j - 1] > t:
                    vs[j], px[j], py[j] = vs[j - 1], px[j - 1], py[j - 1]
                    j -= 1
                vs[j], px[j], py[j] = t, tx, ty
        area = F32(0)  # area() :170-177
        for k in range(n - 2):
            tri = ((px[0] - px[k + 2]) * (py[k + 1] - py[k + 2]) - (py[0] - py[k + 2]) * (px[k + 1] - px[k + 2])) / F32(2)
            area = F32(area + abs(F32(tri)))
        return area


def rotated_iou(rb1, rb2):
    """devRotateIoU, eval/iou.py:394-399."""
    rb1 = np.asarray(rb1, dtype=F32)
    rb2 = np.asarray(rb2, dtype=F32)
    with np.errstate(all="ignore"):
        a1 = rb1[2] * rb1[3]
        a2 = rb2[2] * rb2[3]
        ai = rotated_inter_area(rb1, rb2)
        return F32(ai / (a1 + a2 - ai))


def nms_rotated(dets, thresh):
    """rotate_nms_gpu, eval/iou.py:438-473. dets f32[n,6] (cx,cy,dx,dy,angle,score)."""
    dets = np.asarray(dets, dtype=F32)
    n = dets.shape[0]
    if n == 0:
        return []
    order = _order_desc(dets[:, 5])
    b = dets[order]
    sup = np.zeros((n, n), dtype=bool)
    for i in range(n):
        for j in range(i + 1, n):
            sup[i, j] = rotated_iou(b[i, :5], b[j, :5]) > F32(thresh)
    return [int(order[i]) for i in _greedy(sup)]


# ----------------------------------------------------------------------------
# a6-a9  network (torch CPU fp32 functional restatement)
#        networks/pointpillars8_shared.py:11-60,63-111,114-181,299-343,418-431
# ----------------------------------------------------------------------------
def _t(x):
    import torch
    return torch.from_numpy(np.ascontiguousarray(x))


def pfn(voxels, num_points, coors, sd, setup):
    """PointNet.forward (:30-60): decorate 4->9 features, zero padded rows, Conv1d(9->64,
    no bias) + BatchNorm1d(eval) + ReLU, max over the T slots.  Padded slots contribute
    relu(beta - mean*gamma/sqrt(var+eps)) to the max.  -> f32[P,64]."""
    import torch
    import torch.nn.functional as Fn
    v = _t(np.asarray(voxels, dtype=F32))
    npts = _t(np.asarray(num_points, dtype=np.int32))
    co = _t(np.asarray(coors, dtype=np.int32))
    vx, vy = setup["voxel_size"][0], setup["voxel_size"][1]
    x_off = vx / 2 + setup["offset"][0]
    y_off = vy / 2 + setup["offset"][1]
    mean = v[:, :, :3].sum(dim=1, keepdim=True) / npts.type_as(v).view(-1, 1, 1)
    f_cluster = v[:, :, :3] - mean
    f_center = torch.zeros_like(v[:, :, :2])
    f_center[:, :, 0] = v[:, :, 0] - (co[:, 0].float().unsqueeze(1) * vx + x_off)
    f_center[:, :, 1] = v[:, :, 1] - (co[:, 1].float().unsqueeze(1) * vy + y_off)
    feats = torch.cat([v, f_cluster, f_center], dim=-1)
    t = feats.shape[1]
    mask = (npts.view(-1, 1) > torch.arange(t, dtype=torch.int32).view(1, -1)).unsqueeze(-1).type_as(feats)
    feats = feats * mask
    p = "pillar_point_net.pfn_layers."
    x = Fn.conv1d(feats.permute(0, 2, 1).contiguous(), _t(sd[p + "0.weight"]))
    x = Fn.batch_norm(x, _t(sd[p + "1.running_mean"]), _t(sd[p + "1.running_var"]), _t(sd[p + "1.weight"]),
                      _t(sd[p + "1.bias"]), training=False, eps=1e-5)
    x = torch.relu(x)
    return x.max(dim=2)[0].numpy()


def scatter(feat, coors, grid_size):
    """PointPillarsScatter.forward (:76-111): canvas[c, cx*ny+cy] = feat[p,c] -> [1,64,nx,ny]."""
    nx, ny = int(grid_size[0]), int(grid_size[1])
    canvas = np.zeros((feat.shape[1], nx * ny), dtype=F32)
    idx = coors[:, 0].astype(np.int64) * ny + coors[:, 1].astype(np.int64)
    canvas[:, idx] = np.asarray(feat, dtype=F32).T
    return canvas.reshape(1, feat.shape[1], nx, ny)


def _norm(x, sd, key, norm):
    import torch.nn.functional as Fn
    if norm == "instance":  # InstanceNorm2d(eps=1e-3, affine=False, no running stats) :128
        return Fn.instance_norm(x, eps=1e-3)
    return Fn.batch_norm(x, _t(sd[key + ".running_mean"]), _t(sd[key + ".running_var"]), _t(sd[key + ".weight"]),
                         _t(sd[key + ".bias"]), training=False, eps=1e-3)  # pointpillars8_export.py:65


def backbone(canvas, sd, norm="instance"):
    """RPN.forward (:173-181): three blocks of [conv s2, norm, relu, Resnet2...] with
    pre-activation residual units (:418-431) + three ConvTranspose upsamplers, concat."""
    import torch
    import torch.nn.functional as Fn
    x = _t(np.asarray(canvas, dtype=F32))
    ups = []
    with torch.no_grad():
        for bi, nres in ((1, (1, 0)), (2, (1, 1, 0)), (3, (1, 1, 0))):
            pre = f"rpn.block{bi}."
            x = Fn.conv2d(x, _t(sd[pre + "0.weight"]), stride=2, padding=1)
            x = torch.relu(_norm(x, sd, pre + "1", norm))
            for j, nl in enumerate(nres):
                cb = f"{pre}{3 + j}.conv_block."
                y = torch.relu(_norm(x, sd, cb + "0", norm))
                y = Fn.conv2d(y, _t(sd[cb + "2.weight"]), padding=1)
                if nl == 1:
                    y = torch.relu(_norm(y, sd, cb + "3", norm))
                    y = Fn.conv2d(y, _t(sd[cb + "5.weight"]), padding=1)
                x = x + y
            s = (1, 2, 4)[bi - 1]
            u = Fn.conv_transpose2d(x, _t(sd[f"rpn.deconv{bi}.0.weight"]), stride=s)
            ups.append(torch.relu(_norm(u, sd, f"rpn.deconv{bi}.1", norm)))
        return torch.cat(ups, dim=1).numpy()


def head(rpn_out, sd, num_anchor_per_loc=9):
    """SharedHead.forward (:323-343): three 1x1 convs with bias; outputs ordered
    (anchor type a, x, y[, code]).  -> cls [1,A,1], box [1,A,7], dir [1,A,2]."""
    import torch
    import torch.nn.functional as Fn
    x = _t(np.asarray(rpn_out, dtype=F32))
    with torch.no_grad():
        n = x.shape[0]
        cls = Fn.conv2d(x, _t(sd["heads.conv_cls.weight"]), _t(sd["heads.conv_cls.bias"])).reshape(n, -1, 1)
        box = Fn.conv2d(x, _t(sd["heads.conv_box.weight"]), _t(sd["heads.conv_box.bias"]))
        _, _, h, w = box.shape
        box = box.view(n, num_anchor_per_loc, 7, h, w).permute(0, 1, 3, 4, 2).contiguous().view(n, -1, 7)
        dr = Fn.conv2d(x, _t(sd["heads.conv_dir.weight"]), _t(sd["heads.conv_dir.bias"]))
        dr = dr.view(n, num_anchor_per_loc, 2, h, w).permute(0, 1, 3, 4, 2).contiguous().view(n, -1, 2)
    return cls.numpy(), box.numpy(), dr.numpy()


# ----------------------------------------------------------------------------
# a10  Inference.infer_gpu               framework/inference.py:26-138 (+ :9-24, :689-703)
# ----------------------------------------------------------------------------
NMS_PRE_MAX = 1000
NMS_POST_MAX = 300
NMS_IOU_THR = 0.1
SCORE_THR = 0.05


def sigmoid_f32(x):
    """Correctly rounded fp32 sigmoid (computed in fp64, rounded once).  torch.sigmoid
    (inference.py:51) is within 1 ulp of this; the HIP path computes the same way."""
    x = np.asarray(x, dtype=np.float64)
    return (1.0 / (1.0 + np.exp(-x))).astype(F32)


def postprocess(cls_preds, box_preds, dir_preds, anchors_mask, anchors, class_masks, center_limit,
                nms_mode="aabb", detail=False, nms_fn=None):
    """Per class: anchor mask -> sigmoid -> score >= 0.05 -> top-1000 -> decode -> standup
    AABB -> NMS(0.1) -> first 300 -> direction flip -> range mask (quirk: dims vs upper
    limits, :107-109) -> limit_period(2*pi).  Returns (det f32[k,9] rows
    x,y,z,l,w,h,r,score,class_index , per-class counts).
    detail=True additionally returns, per class, the intermediate selections (test diagnostics: which
    anchors were candidates / kept, so a differing detection can be traced to the decision that flipped):
    dict(idx top-k anchor ids in score order, score, dets NMS input, keep positions into idx (all NMS
    survivors, before the 300 cut), final anchor id per output row).  nms_fn(dets, thr) overrides the NMS
    implementation (the C oracle for large candidate sets)."""
    info = []
    cls_preds = np.asarray(cls_preds, dtype=F32).reshape(-1)
    box_preds = np.asarray(box_preds, dtype=F32).reshape(-1, 7)
    dir_preds = np.asarray(dir_preds, dtype=F32).reshape(-1, 2)
    anchors_mask = np.asarray(anchors_mask).reshape(-1).astype(bool)
    rows, counts = [], []
    lim = np.asarray(center_limit, dtype=np.float64)
    for ci, (name, (s, e)) in enumerate(class_masks.items()):
        idx = np.nonzero(anchors_mask[s:e])[0] + s
        sc = sigmoid_f32(cls_preds[idx])
        keep = sc >= F32(SCORE_THR)
        idx, sc = idx[keep], sc[keep]
        if idx.size == 0:
            counts.append(0)
            info.append(dict(idx=idx, score=sc, dets=np.zeros((0, 5), F32), keep=np.zeros(0, np.int64), final=idx, n_cand=0))
            continue
        n_cand = int(idx.size)
        # topk: score descending, ties by lower anchor index (torch.topk leaves ties unspecified)
        o = np.lexsort((idx, -sc.astype(np.float64)))[:NMS_PRE_MAX]
        idx, sc = idx[o], sc[o]
        dirl = dir_preds[idx, 1] > dir_preds[idx, 0]  # torch.max(dim=-1)[1]: first max wins on ties
        boxes = box_decode(box_preds[idx], anchors[idx])
        if nms_mode == "aabb":
            corners = center_to_corner_box2d(boxes[:, :2], boxes[:, 3:5], boxes[:, 6])
            dets = np.concatenate([corner_to_standup_nd(corners), sc[:, None]], axis=1)
            keep_all = (nms_fn or nms_aabb)(dets, NMS_IOU_THR)
        else:
            dets = np.concatenate([boxes[:, [0, 1, 3, 4, 6]], sc[:, None]], axis=1)
            keep_all = (nms_fn or nms_rotated)(dets, NMS_IOU_THR)
        sel = np.asarray(keep_all[:NMS_POST_MAX], dtype=np.int64)
        b = boxes[sel].copy()
        s_sel = sc[sel]
        opp = (b[:, 6] > 0) ^ dirl[sel]
        b[:, 6] = (b[:, 6].astype(np.float64) + np.where(opp, np.pi, 0.0)).astype(F32)  # :101 (f64 add, f32 store)
        rm = np.any(b[:, :3] > lim[:3], axis=1) & np.any(b[:, 3:6] < lim[3:], axis=1)
        b, s_sel = b[rm], s_sel[rm]
        info.append(dict(idx=idx, score=sc, dets=dets, keep=np.asarray(keep_all, dtype=np.int64), final=idx[sel][rm], n_cand=n_cand))
        b[:, 6] = limit_period(b[:, 6], 0.5, 2 * np.pi)
        rows.append(np.concatenate([b, s_sel[:, None], np.full((b.shape[0], 1), ci, dtype=F32)], axis=1))
        counts.append(int(b.shape[0]))
    det = np.concatenate(rows, axis=0).astype(F32) if rows else np.zeros((0, 9), dtype=F32)
    if detail:
        return det, counts, info
    return det, counts
