"""eval.eval of the reference (eval/eval.py): KITTI-style AP for the lidar frame, `get_official_eval_result`
and the functions under it with the reference's names, arguments and return values.

What runs where:
* rotated BEV overlaps: on the device, ONE `pp_rotated_iou_eval` launch for all frames of a part (eval/iou.py:606-638);
* the height term of the 3-D overlap (d3_box_overlap_kernel_lidar, eval.py:148-170): vectorised numpy in the dtype
  the annos carry, as the reference's numba loop computes it;
* the AP table: the data set is flattened once (`_Scene`: CSR over frames), the ignore codes of a class are one vectorised
  pass over all boxes, and each (class, overlap threshold) cell is ONE native call (`pp_eval_class_ap`: matching at
  threshold 0, the 41 recall-sampled score thresholds, tp/fp/fn per threshold, precision / recall with the running
  maximum) -- no Python loop over frames;
* compute_statistics_jit / fused_compute_statistics (eval.py:62-119,182-216) remain available with the reference's
  signatures on the same native matcher (`pp_eval_statistics`, `pp_eval_fused_statistics`).
The camera-frame variants (calculate_iou_partly_camera, d3_box_overlap_camera) are not on this repository's path
(`frame = 'lidar'` is hard-coded at eval.py:467) and are not provided.
"""
import ctypes

import numpy as np

from .. import _lib
from .iou import rotate_iou_gpu_eval

MIN_OVERLAPS = {'vehicle': [0.7, 0.5], 'pedestrian': [0.5, 0.25], 'cyclist': [0.5, 0.25]}  # eval.py:462-464


def get_range(x, y):
    return np.sqrt(x * x + y * y)


def _ignore_codes(gname, gloc, gnpts, dname, dloc, cls, num_points_thresh, range_thresh):
    """The reference's per-box ignore codes (eval.py:10-39) for any number of boxes at once: -1 other class / no points /
    out of range, 0 counts, 1 too few points (a detection matched to it is not a false positive)."""
    ignored_gt = np.full(gname.shape[0], -1, dtype=np.int64)
    if gname.shape[0]:
        live = (gname == cls) & (gnpts != 0) & (get_range(gloc[:, 0], gloc[:, 1]) < range_thresh)
        ignored_gt[live] = np.where(gnpts[live] > num_points_thresh, 0, 1)
    ignored_dt = np.full(dname.shape[0], -1, dtype=np.int64)
    if dname.shape[0]:
        ignored_dt[(dname == cls) & (get_range(dloc[:, 0], dloc[:, 1]) < range_thresh)] = 0
    return ignored_gt, ignored_dt


def _lower_names(anno):
    return np.char.lower(np.asarray(anno["name"], dtype=str)) if len(anno["name"]) else np.zeros((0,), dtype=str)


def clean_data(gt_anno, dt_anno, current_class, num_points_thresh, range_thresh):
    """eval.py:10-39 for one frame -> (num_valid_gt, ignored_gt list, ignored_dt list)."""
    ig, idt = _ignore_codes(_lower_names(gt_anno), np.asarray(gt_anno["location"]).reshape(-1, 3), np.asarray(gt_anno["num_points"]).reshape(-1),
                            _lower_names(dt_anno), np.asarray(dt_anno["location"]).reshape(-1, 3), current_class.lower(), num_points_thresh,
                            range_thresh)
    return int((ig == 0).sum()), ig.tolist(), idt.tolist()


def get_thresholds(scores, num_gt, num_sample_pts=41):
    """eval.py:42-59 (the score list is sorted in place there; here a copy is)."""
    scores = np.sort(np.asarray(scores))[::-1]
    current_recall = 0
    thresholds = []
    last = len(scores) - 1
    for i, score in enumerate(scores):
        l_recall = (i + 1) / num_gt
        r_recall = (i + 2) / num_gt if i < last else l_recall
        if (r_recall - current_recall) < (current_recall - l_recall) and i < last:
            continue
        thresholds.append(score)
        current_recall += 1 / (num_sample_pts - 1.0)
    return thresholds


def _p(a):
    return ctypes.c_void_p(a.ctypes.data)


def compute_statistics_jit(overlaps, ignored_gt, ignored_det, dt_scores, min_overlap, thresh=0, compute_fp=False):
    """eval.py:62-119 -> (tp, fp, fn, thresholds)."""
    ov = np.ascontiguousarray(overlaps, dtype=np.float64)
    ig = np.ascontiguousarray(ignored_gt, dtype=np.int64)
    idt = np.ascontiguousarray(ignored_det, dtype=np.int64)
    sc = np.ascontiguousarray(dt_scores, dtype=np.float32)
    out = np.zeros(3, dtype=np.int64)
    thr = np.zeros(max(ig.size, 1), dtype=np.float64)
    nthr = np.zeros(1, dtype=np.int64)
    ld = ov.shape[1] if ov.ndim == 2 and ov.shape[1] else max(ig.size, 1)
    _lib.check(_lib.load().pp_eval_statistics(_p(ov), ld, idt.size, ig.size, _p(ig), _p(idt), _p(sc), float(min_overlap), float(thresh),
                                              int(bool(compute_fp)), _p(out), _p(thr), _p(nthr)), None, "pp_eval_statistics")
    return int(out[0]), int(out[1]), int(out[2]), thr[:int(nthr[0])]


def fused_compute_statistics(overlaps, pr, gt_nums, dt_nums, ignored_gts, ignored_dets, dt_scores, min_overlap, thresholds):
    """eval.py:182-216: accumulates (tp, fp, fn) per threshold into pr[:, 0:3] in place."""
    ov = np.ascontiguousarray(overlaps, dtype=np.float64)
    assert pr.dtype == np.float64 and pr.flags.c_contiguous and pr.shape[1] == 4
    g = np.ascontiguousarray(gt_nums, dtype=np.int64)
    d = np.ascontiguousarray(dt_nums, dtype=np.int64)
    ig = np.ascontiguousarray(ignored_gts, dtype=np.int64)
    idt = np.ascontiguousarray(ignored_dets, dtype=np.int64)
    sc = np.ascontiguousarray(dt_scores, dtype=np.float32)
    th = np.ascontiguousarray(thresholds, dtype=np.float64)
    ld = ov.shape[1] if ov.ndim == 2 and ov.shape[1] else 1
    _lib.check(_lib.load().pp_eval_fused_statistics(_p(ov), ld, _p(pr), _p(g), _p(d), int(g.size), _p(ig), _p(idt), _p(sc), float(min_overlap),
                                                    _p(th), int(th.size)), None, "pp_eval_fused_statistics")


def d3_box_overlap_kernel_lidar(boxes, qboxes, rinc, criterion=-1):
    """eval.py:148-170, in place on rinc: BEV intersection x height overlap over the chosen union."""
    if rinc.size == 0:
        return
    top = np.minimum((boxes[:, 2] + boxes[:, 5] / 2)[:, None], (qboxes[:, 2] + qboxes[:, 5] / 2)[None, :])
    bot = np.maximum((boxes[:, 2] - boxes[:, 5] / 2)[:, None], (qboxes[:, 2] - qboxes[:, 5] / 2)[None, :])
    iw = top - bot
    area1 = (boxes[:, 3] * boxes[:, 4] * boxes[:, 5])[:, None]
    area2 = (qboxes[:, 3] * qboxes[:, 4] * qboxes[:, 5])[None, :]
    inc = iw * rinc
    if criterion == -1:
        ua = area1 + area2 - inc
    elif criterion == 0:
        ua = np.broadcast_to(area1, inc.shape)
    elif criterion == 1:
        ua = np.broadcast_to(area2, inc.shape)
    else:
        ua = np.ones_like(inc)
    pos = rinc > 0
    with np.errstate(all="ignore"):
        val = np.where(iw > 0, inc / ua, 0.0)
    rinc[pos] = val[pos].astype(rinc.dtype)


def d3_box_overlap_lidar(boxes, qboxes, criterion=-1):
    rinc = rotate_iou_gpu_eval(boxes[:, [0, 1, 3, 4, 6]], qboxes[:, [0, 1, 3, 4, 6]], 2)
    d3_box_overlap_kernel_lidar(boxes, qboxes, rinc, criterion)
    return rinc


def bev_box_overlap(boxes, qboxes, criterion=-1):
    return rotate_iou_gpu_eval(boxes, qboxes, criterion)


def get_split_parts(num, num_part):
    """eval.py:173-180.  NOTE the reference yields empty leading parts when num < num_part and then fails in
    np.concatenate (:249); here empty parts are simply skipped by the callers."""
    same_part = num // num_part
    remain_num = num % num_part
    if remain_num == 0:
        return [same_part] * num_part
    return [same_part] * num_part + [remain_num]


def _boxes(annos, metric):
    if not annos:
        return np.zeros((0, 5 if metric == 'bev' else 7), dtype=np.float32)
    if metric == 'bev':
        loc = np.concatenate([a["location"][:, :2] for a in annos], 0)
        dims = np.concatenate([a["dimensions"][:, :2] for a in annos], 0)
    else:
        loc = np.concatenate([a["location"] for a in annos], 0)
        dims = np.concatenate([a["dimensions"] for a in annos], 0)
    rots = np.concatenate([a["rotation_y"] for a in annos], 0)
    return np.concatenate([loc, dims, -rots[..., np.newaxis]], axis=1)


def calculate_iou_partly_lidar(gt_annos, dt_annos, metric='bev', num_parts=50):
    """eval.py:238-287: overlaps of every frame, computed part-wise (one device launch per part over the part's
    concatenated boxes) and cut into per-frame blocks."""
    if metric not in ('bev', '3d'):
        raise ValueError("unknown metric")
    num_examples = len(gt_annos)
    split_parts = get_split_parts(num_examples, num_parts)
    total_dt_num = np.array([len(a["name"]) for a in dt_annos], dtype=np.int64)
    total_gt_num = np.array([len(a["name"]) for a in gt_annos], dtype=np.int64)
    parted_overlaps = []
    overlaps = []
    example_idx = 0
    for num_part in split_parts:
        gt_part = gt_annos[example_idx:example_idx + num_part]
        dt_part = dt_annos[example_idx:example_idx + num_part]
        gt_boxes, dt_boxes = _boxes(gt_part, metric), _boxes(dt_part, metric)
        if metric == 'bev':
            part = bev_box_overlap(gt_boxes, dt_boxes).astype(np.float64)
        else:
            part = d3_box_overlap_lidar(gt_boxes, dt_boxes).astype(np.float64)
        parted_overlaps.append(part)
        gi = di = 0
        for i in range(num_part):
            g, d = int(total_gt_num[example_idx + i]), int(total_dt_num[example_idx + i])
            overlaps.append(part[gi:gi + g, di:di + d])
            gi += g
            di += d
        example_idx += num_part
    return overlaps, parted_overlaps, total_gt_num, total_dt_num


class _Scene:
    """All frames of an evaluation as flat arrays (CSR over frames): the ignore codes of a class are then ONE vectorised
    pass over every box of the data set and the AP of a (class, overlap threshold) cell ONE native call
    (pp_eval_class_ap), instead of Python loops over frames around per-frame native calls."""

    def __init__(self, gt_annos, dt_annos):
        assert len(gt_annos) == len(dt_annos)
        self.n = len(gt_annos)
        self.gt_nums = np.array([len(a["name"]) for a in gt_annos], dtype=np.int64)
        self.dt_nums = np.array([len(a["name"]) for a in dt_annos], dtype=np.int64)
        cat = lambda annos, key, tail, dt: (np.concatenate([np.asarray(a[key], dtype=dt).reshape((-1,) + tail) for a in annos], 0)
                                            if annos else np.zeros((0,) + tail, dt))
        self.gt_name = np.concatenate([_lower_names(a) for a in gt_annos]) if self.n else np.zeros((0,), str)
        self.dt_name = np.concatenate([_lower_names(a) for a in dt_annos]) if self.n else np.zeros((0,), str)
        self.gt_loc = cat(gt_annos, "location", (3,), np.float64)
        self.dt_loc = cat(dt_annos, "location", (3,), np.float64)
        self.gt_npts = cat(gt_annos, "num_points", (), np.int64)
        self.dt_score = cat(dt_annos, "score", (), np.float32)
        self._gt_annos, self._dt_annos = gt_annos, dt_annos

    def overlaps(self, metric, num_parts):
        """Part-wise overlap matrices [detections of the part][ground truths of the part] (float64, C order) -- one device
        launch per part, as calculate_iou_partly_lidar(dt_annos, gt_annos, ...) of eval.py:377-383 -- and the frames per part."""
        _, parted, _, _ = calculate_iou_partly_lidar(self._dt_annos, self._gt_annos, metric, num_parts)
        frames = np.array([p for p in get_split_parts(self.n, num_parts)], dtype=np.int64)
        return [np.ascontiguousarray(m, dtype=np.float64) for m in parted], frames

    def class_ap(self, parts, part_frames, cls, min_overlap, num_points_thresh, range_thresh, n_sample_pts=41, codes=None):
        ig, idt = codes if codes is not None else _ignore_codes(self.gt_name, self.gt_loc, self.gt_npts, self.dt_name, self.dt_loc, cls.lower(),
                                                                num_points_thresh, range_thresh)
        prec = np.zeros(n_sample_pts, dtype=np.float64)
        rec = np.zeros(n_sample_pts, dtype=np.float64)
        ptrs = (ctypes.c_void_p * max(len(parts), 1))(*[m.ctypes.data if m.size else None for m in parts])
        _lib.check(_lib.load().pp_eval_class_ap(ptrs, _p(part_frames), len(parts), _p(self.dt_nums), _p(self.gt_nums), self.n, _p(ig), _p(idt),
                                                _p(self.dt_score), float(min_overlap), int((ig == 0).sum()), n_sample_pts, _p(prec), _p(rec)),
                   None, "pp_eval_class_ap")
        return prec, rec


def eval_class_AP(gt_annos, dt_annos, class_names, metric, min_overlaps, frame, num_points_thresh, range_thresh, num_parts=50):
    """Signature and return value of eval.py:363-440: {"recall", "precision"} as [class, overlap threshold, 41 recall samples]."""
    if frame != 'lidar':
        raise ValueError("only the lidar frame is on this repository's path (eval.py:467)")
    scene = _Scene(gt_annos, dt_annos)
    parts, part_frames = scene.overlaps(metric, num_parts)
    n_thr = len(next(iter(min_overlaps.values())))
    precision = np.zeros([len(class_names), n_thr, 41])
    recall = np.zeros([len(class_names), n_thr, 41])
    for m, cls in enumerate(class_names):
        codes = _ignore_codes(scene.gt_name, scene.gt_loc, scene.gt_npts, scene.dt_name, scene.dt_loc, cls.lower(), num_points_thresh, range_thresh)
        for k, min_overlap in enumerate(min_overlaps[cls]):
            precision[m, k], recall[m, k] = scene.class_ap(parts, part_frames, cls, min_overlap, num_points_thresh, range_thresh, codes=codes)
    return {"recall": recall, "precision": precision}


def get_mAP(prec):
    """eval.py:443-447: 11-point interpolation over the 41 sampled recalls."""
    sums = 0
    for i in range(0, prec.shape[-1], 4):
        sums = sums + prec[..., i]
    return sums / 11 * 100


def get_official_eval_result(gt_annos, dt_annos, class_names, range_thresh):
    """eval.py:461-483 -> ([mAP_bev[C,2], mAP_3d[C,2]], report string)."""
    min_overlaps = MIN_OVERLAPS
    metrics = ['bev', '3d']
    frame = 'lidar'
    num_point_threshold = 5
    results = []
    eval_str = ''
    for metric in metrics:
        eval_str += '\n#### Metric: %s, num_points > %d and range < %.2f\n' % (metric, num_point_threshold, range_thresh)
        ret = eval_class_AP(gt_annos, dt_annos, class_names, metric, min_overlaps, frame, num_point_threshold, range_thresh=range_thresh)
        mAP = get_mAP(ret['precision'])
        results.append(mAP)
        for i, cls in enumerate(class_names):
            eval_str += cls + ':\t'
            for j, iou in enumerate(min_overlaps[cls]):
                eval_str += '@%.2f %.4f\t' % (iou, mAP[i][j])
            eval_str += '\n'
    return results, eval_str
