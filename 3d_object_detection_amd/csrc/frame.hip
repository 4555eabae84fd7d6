// pp_infer_frame / pp_infer_batch: the whole hot path on one stream, no host synchronisation
// (train.py:222-237 crosses the host/device boundary >= 3 times and synchronises >= 14 times per frame).
// A batch carries nb independent frames through ONE set of conv/deconv/head launches (grid.z = frame):
// the launches get several rounds of workgroups, so the prologue/epilogue of one round overlaps the MFMA
// phase of the next instead of being exposed once per frame and layer.
#include <cstdlib>
#include <cstring>
#include "pp_common.h"

extern "C" int pp_infer_batch(pp_ctx* ctx, const float* const* pts_h, const int32_t* n_h, int nb, float* det, int32_t* det_count,
                              int nms_mode, void* stream_)
{
    if (!ctx) return PP_E_ARG;
    hipStream_t stream = (hipStream_t)stream_;
    if (!ctx->weights_ready) return pp_fail(ctx, PP_E_STATE, "pp_infer_batch: weights not committed");
    if (ctx->A == 0) return pp_fail(ctx, PP_E_STATE, "pp_infer_batch: call pp_set_anchors first");
    if (!det || !det_count || !pts_h || !n_h) return pp_fail(ctx, PP_E_ARG, "pp_infer_batch: null pointer");
    if (nb < 1 || nb > ctx->max_batch) return pp_fail(ctx, PP_E_ARG, "pp_infer_batch: nb exceeds cfg.max_batch");
    const pp_config& c = ctx->cfg;
    const size_t mv = (size_t)c.max_voxels;
    const size_t vs = mv * c.max_num_points * c.num_point_features;
    const size_t cells = (size_t)ctx->gx * ctx->gy;
    const size_t A = (size_t)ctx->A;
    int rc;
    // PP_BATCH_STAGES=0: the earlier scheme (one launch per stage AND frame, frames dealt round-robin to the
    // caller's stream plus PP_AUX_STREAMS internal ones) -- kept for A/B timing and as the parity reference of the
    // batched stage kernels.
    static const bool batched = !(getenv("PP_BATCH_STAGES") && getenv("PP_BATCH_STAGES")[0] == '0');
    if (batched) {
        // The integer stages are latency-bound at one frame per launch (a few workgroups each); with blockIdx.z =
        // frame every stage is ONE launch per group of <= 16 frames: 21 launches per group instead of 21 per frame.
        if ((rc = pp_build_tables(ctx))) return rc;
        for (int b0 = 0; b0 < nb; b0 += PP_GROUP) {
            const int g = nb - b0 < PP_GROUP ? nb - b0 : PP_GROUP;
            pp_in_group in;
            memset(&in, 0, sizeof(in));
            for (int i = 0; i < g; ++i) { in.pts[i] = pts_h[b0 + i]; in.n[i] = n_h[b0 + i]; }
            if ((rc = pp_stage_mark(ctx, stream, PP_ST_VOXELIZE))) return rc;
            if ((rc = pp_voxelize_group(ctx, b0, g, in, stream))) return rc;
            if ((rc = pp_stage_mark(ctx, stream, PP_ST_MASK))) return rc;
            if ((rc = pp_anchor_mask_group(ctx, b0, g, stream))) return rc;
            if ((rc = pp_stage_mark(ctx, stream, PP_ST_PFN))) return rc;
            if ((rc = pp_pfn_pmap_group(ctx, b0, g, stream))) return rc;
        }
        if ((rc = pp_stage_mark(ctx, stream, PP_ST_CONV))) return rc;
        if ((rc = pp_run_backbone(ctx, nullptr, nb, stream, ctx->f_pmap, ctx->f_feat))) return rc;
        if ((rc = pp_stage_mark(ctx, stream, PP_ST_HEAD))) return rc;
        if ((rc = pp_run_head_fused(ctx, ctx->f_cls, ctx->f_box, ctx->f_dir, nb, stream))) return rc;
        if ((rc = pp_stage_mark(ctx, stream, PP_ST_POST))) return rc;
        for (int b0 = 0; b0 < nb; b0 += PP_GROUP) {
            const int g = nb - b0 < PP_GROUP ? nb - b0 : PP_GROUP;
            if (b0 && (rc = pp_stage_mark(ctx, stream, PP_ST_POST))) return rc;
            if ((rc = pp_postprocess_group(ctx, b0, g, det, det_count, nms_mode, stream))) return rc;
        }
        return pp_stage_mark(ctx, stream, -1);
    }
    // frames are dealt round-robin to (1 + naux) streams: the caller's and naux internal ones.  The GPU exposes
    // 4 hardware queues by default, so more than 3 internal streams only adds queue-switch overhead.
    static const int naux_env = getenv("PP_AUX_STREAMS") ? atoi(getenv("PP_AUX_STREAMS")) : 3;
    const int naux = naux_env < 0 ? 0 : (naux_env > nb - 1 ? nb - 1 : naux_env);
    auto lane_of = [&](int b) { return naux == 0 ? 0 : b % (naux + 1); }; // 0 = caller's stream, k = slot k's stream
    // Fork: the integer stages of the nb frames are independent and latency-bound, so frames 1.. run on the
    // slots' internal streams concurrently with frame 0 (caller's stream); all join before the shared conv launches.
    if (naux > 0) {
        PP_HIP(hipEventRecord(ctx->ev_fork, stream));
        for (int k = 1; k <= naux; ++k) PP_HIP(hipStreamWaitEvent(ctx->slot[k].stream, ctx->ev_fork, 0));
    }
    for (int b = 0; b < nb; ++b) {
        const int ln = lane_of(b);
        hipStream_t st = (ln == 0) ? stream : ctx->slot[ln].stream;
        float* vox = ctx->f_voxels + b * vs;
        int32_t* coors = ctx->f_coors + b * mv * 3;
        int32_t* npts = ctx->f_npts + b * mv;
        int32_t* num = ctx->f_num + b * 4;
        float* feat = ctx->f_feat + b * mv * 64;
        if ((rc = pp_voxelize_slot(ctx, b, pts_h[b], n_h[b], c.num_point_features, vox, coors, npts, num, st))) return rc;
        if ((rc = pp_anchor_mask_slot(ctx, b, coors, num, ctx->f_mask + b * A, st))) return rc;
        if ((rc = pp_pfn(ctx, vox, coors, npts, num, feat, st))) return rc;
        if ((rc = pp_pillar_map(ctx, coors, num, ctx->f_pmap + b * cells, st))) return rc;
    }
    for (int k = 1; k <= naux; ++k) { // join
        PP_HIP(hipEventRecord(ctx->slot[k].ev_pre, ctx->slot[k].stream));
        PP_HIP(hipStreamWaitEvent(stream, ctx->slot[k].ev_pre, 0));
    }
    // sparse BEV: the first conv gathers from (pillar map, PFN rows); no dense canvas, no 164 MB memset per frame
    if ((rc = pp_run_backbone(ctx, nullptr, nb, stream, ctx->f_pmap, ctx->f_feat))) return rc;
    if ((rc = pp_run_head_fused(ctx, ctx->f_cls, ctx->f_box, ctx->f_dir, nb, stream))) return rc;
    const size_t rows = (size_t)c.num_classes * c.nms_post_max;
    if (naux > 0) {
        PP_HIP(hipEventRecord(ctx->ev_mid, stream));
        for (int k = 1; k <= naux; ++k) PP_HIP(hipStreamWaitEvent(ctx->slot[k].stream, ctx->ev_mid, 0));
    }
    for (int b = 0; b < nb; ++b) {
        const int ln = lane_of(b);
        hipStream_t st = (ln == 0) ? stream : ctx->slot[ln].stream;
        if ((rc = pp_postprocess_slot(ctx, b, ctx->f_cls + b * A, ctx->f_box + b * A * 7, ctx->f_dir + b * A * 2, ctx->f_mask + b * A,
                                      det + b * rows * 9, det_count + b * PP_DET_COUNT_STRIDE, nms_mode, st))) return rc;
    }
    for (int k = 1; k <= naux; ++k) {
        PP_HIP(hipEventRecord(ctx->slot[k].ev_post, ctx->slot[k].stream));
        PP_HIP(hipStreamWaitEvent(stream, ctx->slot[k].ev_post, 0));
    }
    return 0;
}

extern "C" int pp_infer_frame(pp_ctx* ctx, const float* pts, int n, float* det, int32_t* det_count, int nms_mode, void* stream)
{
    const float* p1[1] = {pts};
    const int32_t n1[1] = {n};
    return pp_infer_batch(ctx, p1, n1, 1, det, det_count, nms_mode, stream);
}
