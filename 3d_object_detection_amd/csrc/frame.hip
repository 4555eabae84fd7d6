// pp_infer_frame / pp_infer_batch: the whole hot path on one stream, no host synchronisation
// (train.py:222-237 crosses the host/device boundary >= 3 times and synchronises >= 14 times per frame).
// A batch carries nb independent frames through ONE set of conv/deconv/head launches (grid.z = frame):
// the launches get several rounds of workgroups, so the prologue/epilogue of one round overlaps the MFMA
// phase of the next instead of being exposed once per frame and layer.
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
    int rc;
    // The integer stages are latency-bound at one frame per launch (a few workgroups each); with blockIdx.z =
    // frame every stage is ONE launch per group of <= PP_GROUP (32) frames: 21 launches per group instead of 21 per frame.
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

extern "C" int pp_infer_frame(pp_ctx* ctx, const float* pts, int n, float* det, int32_t* det_count, int nms_mode, void* stream)
{
    const float* p1[1] = {pts};
    const int32_t n1[1] = {n};
    return pp_infer_batch(ctx, p1, n1, 1, det, det_count, nms_mode, stream);
}
