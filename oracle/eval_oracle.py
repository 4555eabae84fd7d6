"""CPU restatement of the reference's KITTI-style AP evaluation (eval/eval.py) -- TEST INFRASTRUCTURE ONLY.

Only tests/ may import this module; the product (3d_object_detection_amd/eval/) never does.  Every function
follows the reference function named in its docstring (file:line), with plain Python loops where the reference
uses numba loops.  The rotated overlap comes from oracle/pp_oracle.c (orc_rotated_iou_eval).  Pinned by
tests/golden/eval_ap.npz, which tests/golden/make_goldens.py makes by running the reference's eval.py itself.
"""
import numpy as np

from . import c_oracle as C

MIN_OVERLAPS = {"vehicle": [0.7, 0.5], "pedestrian": [0.5, 0.25], "cyclist": [0.5, 0.25]}  # eval.py:462-464


def clean_data(gt, dt, cls, num_points_thresh, range_thresh):
    """eval.py:10-39."""
    cls = cls.lower()
    ig, idt, valid = [], [], 0
    for i in range(len(gt["name"])):
        if gt["name"][i].lower() != cls:
            ig.append(-1)
        elif gt["num_points"][i] == 0:
            ig.append(-1)
        elif not np.sqrt(gt["location"][i][0] * gt["location"][i][0] + gt["location"][i][1] * gt["location"][i][1]) < range_thresh:
            ig.append(-1)
        elif gt["num_points"][i] > num_points_thresh:
            ig.append(0)
            valid += 1
        else:
            ig.append(1)
    for i in range(len(dt["name"])):
        if dt["name"][i].lower() == cls and np.sqrt(dt["location"][i][0] ** 2 + dt["location"][i][1] ** 2) < range_thresh:
            idt.append(0)
        else:
            idt.append(-1)
    return valid, np.array(ig, np.int64), np.array(idt, np.int64)


def get_thresholds(scores, num_gt, num_sample_pts=41):
    """eval.py:42-59."""
    scores = np.sort(scores)[::-1]
    cur, out = 0, []
    for i, s in enumerate(scores):
        l = (i + 1) / num_gt
        r = (i + 2) / num_gt if i < len(scores) - 1 else l
        if (r - cur) < (cur - l) and i < len(scores) - 1:
            continue
        out.append(s)
        cur += 1 / (num_sample_pts - 1.0)
    return out


def compute_statistics(overlaps, ig, idt, scores, min_overlap, thresh=0.0, compute_fp=False):
    """eval.py:62-119."""
    nd, ng = idt.size, ig.size
    assigned = [False] * nd
    below = [compute_fp and scores[i] < thresh for i in range(nd)]
    NO = -10000000
    tp = fp = fn = 0
    thr = []
    for i in range(ng):
        if ig[i] == -1:
            continue
        det, valid, best = -1, NO, 0
        for j in range(nd):
            if idt[j] == -1 or assigned[j] or below[j]:
                continue
            ov = overlaps[j, i]
            if not compute_fp and ov > min_overlap and scores[j] > valid:
                det, valid = j, scores[j]
            elif compute_fp and ov > min_overlap and ov > best:
                best, det, valid = ov, j, 1
        if valid == NO and ig[i] == 0:
            fn += 1
        elif valid != NO and ig[i] == 1:
            assigned[det] = True
        elif valid != NO:
            tp += 1
            thr.append(scores[det])
            assigned[det] = True
    if compute_fp:
        for i in range(nd):
            if not (assigned[i] or idt[i] == -1 or below[i]):
                fp += 1
    return tp, fp, fn, np.array(thr, dtype=np.float64)


def frame_overlaps(a, b, metric):
    """calculate_iou_partly_lidar for one frame (eval.py:248-266): rows = boxes of `a`, columns = boxes of `b`;
    d3_box_overlap_lidar + kernel (eval.py:226-230,148-170)."""
    if metric == "bev":
        ab = np.concatenate([a["location"][:, :2], a["dimensions"][:, :2], -a["rotation_y"][:, None]], axis=1)
        bb = np.concatenate([b["location"][:, :2], b["dimensions"][:, :2], -b["rotation_y"][:, None]], axis=1)
        return C.rotated_iou_eval(ab, bb, -1).astype(np.float64)
    ab = np.concatenate([a["location"], a["dimensions"], -a["rotation_y"][:, None]], axis=1)
    bb = np.concatenate([b["location"], b["dimensions"], -b["rotation_y"][:, None]], axis=1)
    rinc = C.rotated_iou_eval(ab[:, [0, 1, 3, 4, 6]], bb[:, [0, 1, 3, 4, 6]], 2)
    for i in range(ab.shape[0]):
        for j in range(bb.shape[0]):
            if rinc[i, j] > 0:
                iw = (min(ab[i, 2] + ab[i, 5] / 2, bb[j, 2] + bb[j, 5] / 2) - max(ab[i, 2] - ab[i, 5] / 2, bb[j, 2] - bb[j, 5] / 2))
                if iw > 0:
                    a1 = ab[i, 3] * ab[i, 4] * ab[i, 5]
                    a2 = bb[j, 3] * bb[j, 4] * bb[j, 5]
                    inc = iw * rinc[i, j]
                    rinc[i, j] = inc / (a1 + a2 - inc)
                else:
                    rinc[i, j] = 0.0
    return rinc.astype(np.float64)


def eval_class_ap(gts, dts, classes, metric, min_overlaps, num_points_thresh, range_thresh):
    """eval.py:363-440 (lidar frame).  The reference's split into 50 parts only batches the overlap kernel; the
    statistics are per frame either way."""
    n = len(gts)
    ovs = [frame_overlaps(dts[i], gts[i], metric) for i in range(n)]  # called with (dt, gt): rows = dt, cols = gt
    nmin = len(list(min_overlaps.values())[0])
    precision = np.zeros([len(classes), nmin, 41])
    recall = np.zeros([len(classes), nmin, 41])
    for m, cls in enumerate(classes):
        prep = [clean_data(gts[i], dts[i], cls, num_points_thresh, range_thresh) for i in range(n)]
        total_valid = sum(p[0] for p in prep)
        scores = [dts[i]["score"].astype("float32") for i in range(n)]
        for k, mo in enumerate(min_overlaps[cls]):
            all_thr = []
            for i in range(n):
                all_thr += compute_statistics(ovs[i], prep[i][1], prep[i][2], scores[i], mo, 0.0, False)[3].tolist()
            thresholds = np.array(get_thresholds(np.array(all_thr), total_valid))
            pr = np.zeros([len(thresholds), 4])
            for i in range(n):
                for t, th in enumerate(thresholds):
                    tp, fp, fn, _ = compute_statistics(ovs[i], prep[i][1], prep[i][2], scores[i], mo, th, True)
                    pr[t, 0] += tp
                    pr[t, 1] += fp
                    pr[t, 2] += fn
            with np.errstate(all="ignore"):
                for i in range(len(thresholds)):
                    recall[m, k, i] = pr[i, 0] / (pr[i, 0] + pr[i, 2])
                    precision[m, k, i] = pr[i, 0] / (pr[i, 0] + pr[i, 1])
            for i in range(len(thresholds)):
                precision[m, k, i] = np.max(precision[m, k, i:], axis=-1)
    return {"recall": recall, "precision": precision}


def get_map(prec):
    """eval.py:443-447."""
    s = 0
    for i in range(0, prec.shape[-1], 4):
        s = s + prec[..., i]
    return s / 11 * 100


def official_result(gts, dts, classes, range_thresh):
    """eval.py:461-483."""
    results, text = [], ""
    for metric in ("bev", "3d"):
        text += "\n#### Metric: %s, num_points > %d and range < %.2f\n" % (metric, 5, range_thresh)
        ret = eval_class_ap(gts, dts, classes, metric, MIN_OVERLAPS, 5, range_thresh)
        m = get_map(ret["precision"])
        results.append(m)
        for i, cls in enumerate(classes):
            text += cls + ":\t"
            for j, iou in enumerate(MIN_OVERLAPS[cls]):
                text += "@%.2f %.4f\t" % (iou, m[i][j])
            text += "\n"
    return results, text
