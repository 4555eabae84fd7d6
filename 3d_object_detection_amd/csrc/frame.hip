// pp_infer_frame: the whole hot path for one cloud on one stream, no host synchronisation
// (train.py:222-237 crosses the host/device boundary >= 3 times and synchronises >= 14 times per frame).
#include "pp_common.h"

extern "C" int pp_infer_frame(pp_ctx* ctx, const float* pts, int n, float* det, int32_t* det_count, int nms_mode, void* stream_)
{
    if (!ctx) return PP_E_ARG;
    hipStream_t stream = (hipStream_t)stream_;
    if (!ctx->weights_ready) return pp_fail(ctx, PP_E_STATE, "pp_infer_frame: weights not committed");
    if (ctx->A == 0) return pp_fail(ctx, PP_E_STATE, "pp_infer_frame: call pp_set_anchors first");
    if (!det || !det_count) return pp_fail(ctx, PP_E_ARG, "pp_infer_frame: null pointer");
    int rc;
    if ((rc = pp_voxelize(ctx, pts, n, ctx->cfg.num_point_features, ctx->f_voxels, ctx->f_coors, ctx->f_npts, ctx->f_num, stream))) return rc;
    if ((rc = pp_anchor_mask(ctx, ctx->f_coors, ctx->f_num, ctx->f_mask, stream))) return rc;
    if ((rc = pp_pfn(ctx, ctx->f_voxels, ctx->f_coors, ctx->f_npts, ctx->f_num, ctx->f_feat, stream))) return rc;
    if ((rc = pp_scatter(ctx, ctx->f_feat, ctx->f_coors, ctx->f_num, ctx->f_canvas, stream))) return rc;
    if ((rc = pp_run_backbone(ctx, ctx->f_canvas, stream))) return rc;
    if ((rc = pp_run_head_fused(ctx, ctx->f_cls, ctx->f_box, ctx->f_dir, stream))) return rc;
    return pp_postprocess(ctx, ctx->f_cls, ctx->f_box, ctx->f_dir, ctx->f_mask, det, det_count, nms_mode, stream);
}
