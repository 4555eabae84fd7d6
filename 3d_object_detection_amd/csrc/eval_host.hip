// Host side of the KITTI-style AP evaluation: the greedy detection <-> ground-truth matching the
// reference runs as numba loops (eval/eval.py:62-119 compute_statistics_jit, :182-216 fused_compute_statistics).
// Plain C++ on host pointers -- no GPU work; lives in libpp_hip.so so the Python mirror (eval/eval.py of this
// package) has one native library.  Arithmetic as numba does it: overlaps float64, scores float32 promoted to
// float64 in comparisons.
#include <algorithm>
#include <cmath>
#include "pp_common.h"

namespace {

struct StatOut { int64_t tp = 0, fp = 0, fn = 0; };

StatOut statistics(const double* ov, int64_t ld, int nd, int ng, const int64_t* ig, const int64_t* idt, const float* sc, double min_overlap,
                   double thresh, bool compute_fp, double* thr_out, int64_t* n_thr)
{
    std::vector<char> assigned(nd, 0), below(nd, 0);
    if (compute_fp)
        for (int i = 0; i < nd; ++i) below[i] = (double)sc[i] < thresh;
    constexpr double NO_DETECTION = -10000000.0;
    StatOut r;
    int64_t nt = 0;
    for (int i = 0; i < ng; ++i) {
        if (ig[i] == -1) continue;
        int det = -1;
        double valid = NO_DETECTION, best = 0.0;
        for (int j = 0; j < nd; ++j) {
            if (idt[j] == -1 || assigned[j] || below[j]) continue;
            const double o = ov[(int64_t)j * ld + i];
            const double s = (double)sc[j];
            if (!compute_fp && o > min_overlap && s > valid) { det = j; valid = s; }
            else if (compute_fp && o > min_overlap && o > best) { best = o; det = j; valid = 1.0; }
        }
        if (valid == NO_DETECTION && ig[i] == 0) ++r.fn;
        else if (valid != NO_DETECTION && ig[i] == 1) assigned[det] = 1;
        else if (valid != NO_DETECTION) {
            ++r.tp;
            if (thr_out) thr_out[nt] = (double)sc[det];
            ++nt;
            assigned[det] = 1;
        }
    }
    if (compute_fp)
        for (int i = 0; i < nd; ++i)
            if (!(assigned[i] || idt[i] == -1 || below[i])) ++r.fp;
    if (n_thr) *n_thr = nt;
    return r;
}

} // namespace

extern "C" int pp_eval_statistics(const double* overlaps, int64_t ov_ld, int det_size, int gt_size, const int64_t* ignored_gt,
                                  const int64_t* ignored_det, const float* dt_scores, double min_overlap, double thresh, int compute_fp,
                                  int64_t* tp_fp_fn, double* thresholds_out, int64_t* n_thresholds)
{
    if (det_size < 0 || gt_size < 0 || !tp_fp_fn) return PP_E_ARG;
    if ((det_size > 0 && (!ignored_det || !dt_scores)) || (gt_size > 0 && !ignored_gt)) return PP_E_ARG;
    if (det_size > 0 && gt_size > 0 && (!overlaps || ov_ld < gt_size)) return PP_E_ARG;
    const StatOut r = statistics(overlaps, ov_ld, det_size, gt_size, ignored_gt, ignored_det, dt_scores, min_overlap, thresh, compute_fp != 0,
                                 thresholds_out, n_thresholds);
    tp_fp_fn[0] = r.tp; tp_fp_fn[1] = r.fp; tp_fp_fn[2] = r.fn;
    return 0;
}

// One part of frames whose overlaps sit in one [sum dt][sum gt] block matrix (the reference computes the overlaps of
// 1/50 of the frames at once and walks the diagonal blocks, eval.py:193-216).
extern "C" int pp_eval_fused_statistics(const double* overlaps, int64_t ov_ld, double* pr, const int64_t* gt_nums, const int64_t* dt_nums,
                                        int n_frames, const int64_t* ignored_gts, const int64_t* ignored_dets, const float* dt_scores,
                                        double min_overlap, const double* thresholds, int n_thresholds)
{
    if (n_frames < 0 || n_thresholds < 0 || (n_thresholds > 0 && (!pr || !thresholds))) return PP_E_ARG;
    if (n_frames > 0 && (!gt_nums || !dt_nums)) return PP_E_ARG;
    int64_t g0 = 0, d0 = 0;
    for (int f = 0; f < n_frames; ++f) {
        const int ng = (int)gt_nums[f], nd = (int)dt_nums[f];
        if (ng < 0 || nd < 0) return PP_E_ARG;
        for (int t = 0; t < n_thresholds; ++t) {
            const StatOut r = statistics(overlaps + d0 * ov_ld + g0, ov_ld, nd, ng, ignored_gts + g0, ignored_dets + d0, dt_scores + d0, min_overlap,
                                         thresholds[t], true, nullptr, nullptr);
            pr[4 * t + 0] += (double)r.tp;
            pr[4 * t + 1] += (double)r.fp;
            pr[4 * t + 2] += (double)r.fn;
        }
        g0 += ng;
        d0 += nd;
    }
    return 0;
}


// One (class, overlap threshold) cell of the AP table in ONE call (replaces the loop nest of eval.py:396-434: per-frame
// compute_statistics_jit calls to collect the true positives' scores, get_thresholds over them, fused_compute_statistics
// per part and the precision / recall arithmetic with its running maximum).
//   parts_h[j]      row-major overlap matrix of part j: [detections of the part][ground truths of the part]
//   part_frames_h   frames per part (sums to n_frames); frames with no boxes are fine
//   dt_nums / gt_nums per frame; ignored_* and dt_scores are concatenated over all frames in frame order
//   precision_h / recall_h: n_sample_pts doubles each, zero beyond the thresholds found (as np.zeros in the reference)
extern "C" int pp_eval_class_ap(const double* const* parts_h, const int64_t* part_frames_h, int n_parts, const int64_t* dt_nums_h,
                                const int64_t* gt_nums_h, int64_t n_frames, const int64_t* ignored_gt_h, const int64_t* ignored_dt_h,
                                const float* dt_scores_h, double min_overlap, int64_t num_valid_gt, int n_sample_pts, double* precision_h,
                                double* recall_h)
{
    if (n_parts < 0 || n_frames < 0 || n_sample_pts < 2 || !precision_h || !recall_h) return PP_E_ARG;
    if (n_frames > 0 && (!parts_h || !part_frames_h || !dt_nums_h || !gt_nums_h)) return PP_E_ARG;
    for (int i = 0; i < n_sample_pts; ++i) precision_h[i] = recall_h[i] = 0.0;
    // geometry of every frame inside its part's block matrix
    struct Fr { const double* ov; int64_t ld; int nd, ng; int64_t d0, g0; };
    std::vector<Fr> fr;
    fr.reserve((size_t)n_frames);
    int64_t f = 0, dsum = 0, gsum = 0;
    for (int j = 0; j < n_parts; ++j) {
        int64_t pg = 0;
        for (int64_t i = 0; i < part_frames_h[j]; ++i) pg += gt_nums_h[f + i];
        int64_t pd0 = 0, pg0 = 0;
        for (int64_t i = 0; i < part_frames_h[j]; ++i, ++f) {
            if (f >= n_frames) return PP_E_ARG;
            const int nd = (int)dt_nums_h[f], ng = (int)gt_nums_h[f];
            if (nd < 0 || ng < 0 || (nd > 0 && ng > 0 && !parts_h[j])) return PP_E_ARG;
            fr.push_back(Fr{parts_h[j] ? parts_h[j] + pd0 * pg + pg0 : nullptr, pg, nd, ng, dsum, gsum});
            pd0 += nd; pg0 += ng; dsum += nd; gsum += ng;
        }
    }
    if (f != n_frames) return PP_E_ARG;
    // pass 1: scores of the matched detections at threshold 0 (eval.py:398-409)
    std::vector<double> scores;
    std::vector<double> tmp;
    for (const Fr& q : fr) {
        tmp.resize((size_t)std::max(q.ng, 1));
        int64_t nt = 0;
        statistics(q.ov, q.ld, q.nd, q.ng, ignored_gt_h + q.g0, ignored_dt_h + q.d0, dt_scores_h + q.d0, min_overlap, 0.0, false, tmp.data(), &nt);
        scores.insert(scores.end(), tmp.begin(), tmp.begin() + nt);
    }
    // get_thresholds (eval.py:42-59): the scores at n_sample_pts evenly spaced recall positions
    std::sort(scores.begin(), scores.end(), [](double a, double b) { return a > b; });
    std::vector<double> thr;
    double current_recall = 0.0;
    const size_t ns = scores.size();
    for (size_t i = 0; i < ns; ++i) {
        const double l_recall = (double)(i + 1) / (double)num_valid_gt;
        const double r_recall = (i + 1 < ns) ? (double)(i + 2) / (double)num_valid_gt : l_recall;
        if ((r_recall - current_recall) < (current_recall - l_recall) && i + 1 < ns) continue;
        thr.push_back(scores[i]);
        current_recall += 1.0 / ((double)n_sample_pts - 1.0);
    }
    if ((int)thr.size() > n_sample_pts) return PP_E_ARG; // cannot happen for consistent inputs (num_valid_gt >= matches)
    // pass 2: tp / fp / fn at every threshold (eval.py:411-429), then precision / recall with the running maximum (:430-434)
    std::vector<double> tp(thr.size(), 0.0), fp(thr.size(), 0.0), fn(thr.size(), 0.0);
    for (const Fr& q : fr)
        for (size_t t = 0; t < thr.size(); ++t) {
            const StatOut r = statistics(q.ov, q.ld, q.nd, q.ng, ignored_gt_h + q.g0, ignored_dt_h + q.d0, dt_scores_h + q.d0, min_overlap, thr[t], true,
                                         nullptr, nullptr);
            tp[t] += (double)r.tp; fp[t] += (double)r.fp; fn[t] += (double)r.fn;
        }
    for (size_t t = 0; t < thr.size(); ++t) {
        recall_h[t] = tp[t] / (tp[t] + fn[t]);     // 0/0 -> nan, as numpy gives the reference
        precision_h[t] = tp[t] / (tp[t] + fp[t]);
    }
    for (size_t t = 0; t < thr.size(); ++t) {      // np.max(precision[i:]) over ALL n_sample_pts entries (zeros behind the last threshold);
        double m = precision_h[t];                  // np.max propagates nan
        bool nan = m != m;
        for (int u = (int)t + 1; u < n_sample_pts && !nan; ++u) {
            if (precision_h[u] != precision_h[u]) nan = true;
            else if (precision_h[u] > m) m = precision_h[u];
        }
        precision_h[t] = nan ? std::nan("") : m;
    }
    return 0;
}
