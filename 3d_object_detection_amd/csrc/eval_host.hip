// Host side of the KITTI-style AP evaluation: the greedy detection <-> ground-truth matching the
// reference runs as numba loops (eval/eval.py:62-119 compute_statistics_jit, :182-216 fused_compute_statistics).
// Plain C++ on host pointers -- no GPU work; lives in libpp_hip.so so the Python mirror (eval/eval.py of this
// package) has one native library.  Arithmetic as numba does it: overlaps float64, scores float32 promoted to
// float64 in comparisons.
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
