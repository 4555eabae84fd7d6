// Context lifetime, weight intake, anchor tables and error reporting of libpp_hip.so.
#include <cmath>
#include <cstdio>
#include <cstring>
#include "pp_common.h"

static std::string g_create_err;

int pp_fail_hip(pp_ctx* ctx, hipError_t e, const char* what, const char* file, int line)
{
    char buf[512];
    snprintf(buf, sizeof(buf), "%s failed: %s (%s:%d)", what, hipGetErrorString(e), file, line);
    if (ctx) ctx->err = buf; else g_create_err = buf;
    (void)hipGetLastError();
    return -(int)e;
}

int pp_fail(pp_ctx* ctx, int code, const char* msg)
{
    if (ctx) ctx->err = msg; else g_create_err = msg;
    return code;
}

extern "C" const char* pp_last_error(pp_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_err.c_str(); }
extern "C" int pp_version(void) { return 3; } // round 3: launch plan keyed on max_batch, 16-bit conv kernels, pp_select_candidates

int pp_stage_mark(pp_ctx* ctx, hipStream_t stream, int id)
{
    if (!ctx->stage_on) return 0;
    if (ctx->stage_used >= ctx->stage_ev.size()) {
        hipEvent_t e;
        PP_HIP(hipEventCreate(&e));
        ctx->stage_ev.push_back(e);
        ctx->stage_id.push_back(-1);
        ctx->stage_stream.push_back(nullptr);
    }
    const hipError_t er = hipEventRecord(ctx->stage_ev[ctx->stage_used], stream);
    if (er != hipSuccess) { ctx->stage_on = false; ctx->stage_used = 0; return pp_fail_hip(ctx, er, "hipEventRecord(stage mark)", __FILE__, __LINE__); }
    ctx->stage_id[ctx->stage_used] = id;
    ctx->stage_stream[ctx->stage_used] = stream;
    ctx->stage_used++;
    return 0;
}

extern "C" int pp_stage_profile_begin(pp_ctx* ctx)
{
    if (!ctx) return PP_E_ARG;
    ctx->stage_on = true;
    ctx->stage_used = 0;
    return 0;
}

extern "C" int pp_stage_profile_end(pp_ctx* ctx, double* ms_h)
{
    if (!ctx || !ms_h) return PP_E_ARG;
    ctx->stage_on = false;
    const size_t used = ctx->stage_used;
    ctx->stage_used = 0; // whatever happens below, the next begin starts clean
    for (int i = 0; i < PP_ST_COUNT; ++i) ms_h[i] = 0.0;
    for (size_t i = 0; i + 1 < used; ++i) {
        const int id = ctx->stage_id[i];
        if (id < 0 || id >= PP_ST_COUNT) continue;
        // an interval is meaningful only between two marks of ONE stream
        if (ctx->stage_stream[i] != ctx->stage_stream[i + 1]) continue;
        PP_HIP(hipEventSynchronize(ctx->stage_ev[i]));
        PP_HIP(hipEventSynchronize(ctx->stage_ev[i + 1]));
        float ms = 0.f;
        PP_HIP(hipEventElapsedTime(&ms, ctx->stage_ev[i], ctx->stage_ev[i + 1]));
        ms_h[id] += ms;
    }
    return 0;
}

template <typename T>
static hipError_t dalloc(T** p, size_t count)
{
    return hipMalloc((void**)p, count * sizeof(T) + 256);
}

static int create_impl(pp_ctx* ctx)
{
    const pp_config& c = ctx->cfg;
    PP_HIP(hipSetDevice(ctx->device));
    size_t cells = (size_t)c.grid_size[0] * c.grid_size[1] * c.grid_size[2];
    size_t mp = (size_t)c.max_points;
    ctx->slot.resize(ctx->max_batch);
    for (int b = 0; b < ctx->max_batch; ++b) {
        pp_slot& S = ctx->slot[b];
        // cell_first | slots | vox_scalars share one allocation: one 0x7F fill per frame covers all three
        const size_t nslots = (size_t)c.max_voxels * c.max_num_points;
        PP_HIP(dalloc(&S.cell_first, cells + nslots + 4));
        S.slots = S.cell_first + cells;
        S.vox_scalars = S.slots + nslots;
        PP_HIP(dalloc(&S.pt_cell, mp));
        PP_HIP(dalloc(&S.pt_rank, mp));
        PP_HIP(dalloc(&S.wave_cnt, mp / 64 + 8));
        PP_HIP(dalloc(&S.occ, (size_t)ctx->gx * ctx->gy));
    }
    PP_HIP(hipMalloc((void**)&ctx->d_pre, sizeof(pp_pre_frame) * ctx->max_batch));
    PP_HIP(hipMalloc((void**)&ctx->d_post, sizeof(pp_post_frame) * ctx->max_batch));
    PP_HIP(dalloc(&ctx->pfn_w, 9 * 64));
    PP_HIP(dalloc(&ctx->pfn_scale, 64));
    PP_HIP(dalloc(&ctx->pfn_shift, 64));
    // frame buffers for the fused path
    size_t mv = (size_t)c.max_voxels * ctx->max_batch; // per-frame buffers are stored [max_batch][...]
    PP_HIP(dalloc(&ctx->f_voxels, mv * c.max_num_points * c.num_point_features));
    PP_HIP(dalloc(&ctx->f_coors, mv * 3));
    PP_HIP(dalloc(&ctx->f_npts, mv));
    PP_HIP(dalloc(&ctx->f_num, 4 * (size_t)ctx->max_batch));
    PP_HIP(dalloc(&ctx->f_feat, mv * 64));
    PP_HIP(dalloc(&ctx->f_canvas, (size_t)64 * ctx->gx * ctx->gy)); // dense canvas: stand-alone pp_scatter/pp_backbone only
    PP_HIP(dalloc(&ctx->f_pmap, (size_t)ctx->max_batch * ctx->gx * ctx->gy));
    size_t HW = (size_t)ctx->H * ctx->W;
    size_t Amax = HW * c.num_anchor_per_loc * ctx->max_batch;
    PP_HIP(dalloc(&ctx->f_mask, Amax));
    PP_HIP(dalloc(&ctx->f_cls, Amax));
    PP_HIP(dalloc(&ctx->f_box, Amax * 7));
    PP_HIP(dalloc(&ctx->f_dir, Amax * 2));
    int rc = pp_net_create(ctx);
    if (rc) return rc;
    rc = pp_post_create(ctx);
    return rc;
}

extern "C" pp_ctx* pp_create(int device, const pp_config* cfg)
{
    if (!cfg) { pp_fail(nullptr, PP_E_ARG, "pp_create: null config"); return nullptr; }
    if (cfg->grid_size[2] != 1 || cfg->grid_size[0] < 2 || cfg->grid_size[1] < 2 || (cfg->grid_size[0] & 1) ||
        (cfg->grid_size[1] & 1)) {
        pp_fail(nullptr, PP_E_ARG, "pp_create: grid must be [gx,gy,1] with even gx,gy (multiples of 8 for the backbone)");
        return nullptr;
    }
    if (cfg->max_voxels <= 0 || cfg->max_num_points <= 0 || cfg->max_points <= 0 || cfg->num_classes <= 0 ||
        cfg->num_classes > PP_MAX_CLASSES || cfg->nms_pre_max <= 0 || cfg->nms_pre_max > 4096 ||
        cfg->nms_post_max <= 0 || cfg->nms_post_max > cfg->nms_pre_max || cfg->nms_post_max > 1024 || cfg->max_batch < 0 ||
        cfg->max_batch > 64) {
        pp_fail(nullptr, PP_E_ARG, "pp_create: size out of range");
        return nullptr;
    }
    pp_ctx* ctx = new pp_ctx();
    ctx->cfg = *cfg;
    ctx->max_batch = cfg->max_batch > 0 ? cfg->max_batch : 1;
    ctx->cfg.max_batch = ctx->max_batch;
    ctx->device = device;
    ctx->gx = cfg->grid_size[0];
    ctx->gy = cfg->grid_size[1];
    ctx->H = ctx->gx / 2;
    ctx->W = ctx->gy / 2;
    int rc = create_impl(ctx);
    if (rc) {
        g_create_err = ctx->err;
        pp_destroy(ctx);
        return nullptr;
    }
    return ctx;
}

// Device tables of the per-frame buffers the batched integer stages index with blockIdx.z.  The mask / head
// pointers depend on the anchor count, so the tables are (re)built on the first batch after pp_set_anchors.
int pp_build_tables(pp_ctx* ctx)
{
    if (ctx->tab_A == ctx->A) return 0;
    const pp_config& c = ctx->cfg;
    const size_t mv = (size_t)c.max_voxels;
    const size_t vs = mv * c.max_num_points * c.num_point_features;
    const size_t cells = (size_t)ctx->gx * ctx->gy;
    const size_t A = (size_t)ctx->A;
    std::vector<pp_pre_frame> pre(ctx->max_batch);
    std::vector<pp_post_frame> post(ctx->max_batch);
    for (int b = 0; b < ctx->max_batch; ++b) {
        const pp_slot& S = ctx->slot[b];
        pp_pre_frame& f = pre[b];
        PP_SET(f.pt_cell, S.pt_cell); PP_SET(f.cell_first, S.cell_first); PP_SET(f.wave_cnt, S.wave_cnt); PP_SET(f.pt_rank, S.pt_rank);
        PP_SET(f.slots, S.slots); PP_SET(f.scalars, S.vox_scalars); PP_SET(f.occ, S.occ);
        PP_SET(f.voxels, ctx->f_voxels + b * vs);
        PP_SET(f.coors, ctx->f_coors + b * mv * 3);
        PP_SET(f.npts, ctx->f_npts + b * mv);
        PP_SET(f.num, ctx->f_num + b * 4);
        PP_SET(f.mask, ctx->f_mask + b * A);
        PP_SET(f.feat, ctx->f_feat + b * mv * 64);
        PP_SET(f.pmap, ctx->f_pmap + b * cells);
        pp_post_frame& q = post[b];
        PP_SET(q.cls, ctx->f_cls + b * A); PP_SET(q.box, ctx->f_box + b * A * 7); PP_SET(q.dir, ctx->f_dir + b * A * 2); PP_SET(q.mask, ctx->f_mask + b * A);
        pp_post_fill_table(ctx, b, &q);
    }
    PP_HIP(hipMemcpy(ctx->d_pre, pre.data(), sizeof(pp_pre_frame) * pre.size(), hipMemcpyHostToDevice));
    PP_HIP(hipMemcpy(ctx->d_post, post.data(), sizeof(pp_post_frame) * post.size(), hipMemcpyHostToDevice));
    ctx->tab_A = ctx->A;
    return 0;
}

extern "C" void pp_destroy(pp_ctx* ctx)
{
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    pp_net_destroy(ctx);
    pp_post_destroy(ctx);
    for (pp_slot& S : ctx->slot) {
        void* sp[] = {S.cell_first, S.pt_cell, S.pt_rank, S.wave_cnt, S.occ};
        for (void* q : sp)
            if (q) (void)hipFree(q);
    }
    if (ctx->d_pre) (void)hipFree(ctx->d_pre);
    if (ctx->d_post) (void)hipFree(ctx->d_post);
    void* ptrs[] = {ctx->anchors, ctx->rect_x, ctx->rect_y, ctx->rects, ctx->pfn_w, ctx->pfn_scale, ctx->pfn_shift,
                    ctx->f_voxels, ctx->f_coors, ctx->f_npts, ctx->f_num, ctx->f_feat, ctx->f_canvas, ctx->f_mask, ctx->f_pmap,
                    ctx->f_cls, ctx->f_box, ctx->f_dir};
    for (void* p : ptrs)
        if (p) (void)hipFree(p);
    for (hipEvent_t e : ctx->prof_ev) (void)hipEventDestroy(e);
    for (hipEvent_t e : ctx->stage_ev) (void)hipEventDestroy(e);
    delete ctx;
}

extern "C" int pp_set_precision(pp_ctx* ctx, int mode)
{
    if (!ctx || mode < 0 || mode > 4) return pp_fail(ctx, PP_E_ARG, "pp_set_precision: mode must be 0 (fp32), 1 (bf16x3), 2 (bf16), 3 (fp16) or 4 (fp16s)");
    if (ctx->precision != mode) ctx->weights_ready = false; // the tilings and weight images are chosen at pp_commit_weights
    ctx->precision = mode;
    return 0;
}

extern "C" int pp_load_weights(pp_ctx* ctx, const char* name, const void* host_ptr, const int64_t* shape, int ndim)
{
    if (!ctx || !name || !host_ptr || !shape || ndim < 0 || ndim > 4) return pp_fail(ctx, PP_E_ARG, "pp_load_weights: bad argument");
    pp_tensor_h t;
    int64_t n = 1;
    for (int i = 0; i < ndim; ++i) { t.shape.push_back(shape[i]); n *= shape[i]; }
    if (n <= 0 || n > (1 << 26)) return pp_fail(ctx, PP_E_ARG, "pp_load_weights: bad shape");
    t.data.assign((const float*)host_ptr, (const float*)host_ptr + n);
    ctx->host_w[name] = std::move(t);
    ctx->weights_ready = false;
    return 0;
}

static const pp_tensor_h* find_w(pp_ctx* ctx, const std::string& k, int64_t numel)
{
    auto it = ctx->host_w.find(k);
    if (it == ctx->host_w.end() || (int64_t)it->second.data.size() != numel) return nullptr;
    return &it->second;
}

extern "C" int pp_commit_weights(pp_ctx* ctx)
{
    if (!ctx) return PP_E_ARG;
    PP_HIP(hipSetDevice(ctx->device));
    // PFN: Conv1d weight [64,9,1] -> [9][64]; BatchNorm1d(eval, eps 1e-5) -> scale/shift
    const std::string p = "pillar_point_net.pfn_layers.";
    const pp_tensor_h *w = find_w(ctx, p + "0.weight", 64 * 9), *g = find_w(ctx, p + "1.weight", 64),
                      *b = find_w(ctx, p + "1.bias", 64), *rm = find_w(ctx, p + "1.running_mean", 64),
                      *rv = find_w(ctx, p + "1.running_var", 64);
    if (!w || !g || !b || !rm || !rv) return pp_fail(ctx, PP_E_NAME, "pp_commit_weights: PFN tensors missing or mis-shaped");
    float wT[9 * 64], sc[64], sh[64];
    for (int c = 0; c < 64; ++c) {
        for (int k = 0; k < 9; ++k) wT[k * 64 + c] = w->data[c * 9 + k];
        double s = (double)g->data[c] / std::sqrt((double)rv->data[c] + 1e-5);
        sc[c] = (float)s;
        sh[c] = (float)((double)b->data[c] - (double)rm->data[c] * s);
    }
    PP_HIP(hipMemcpy(ctx->pfn_w, wT, sizeof(wT), hipMemcpyHostToDevice));
    PP_HIP(hipMemcpy(ctx->pfn_scale, sc, sizeof(sc), hipMemcpyHostToDevice));
    PP_HIP(hipMemcpy(ctx->pfn_shift, sh, sizeof(sh), hipMemcpyHostToDevice));
    int rc = pp_net_commit(ctx);
    if (rc) return rc;
    ctx->weights_ready = true;
    return 0;
}

extern "C" int pp_set_anchors(pp_ctx* ctx, const float* anchors_h, const int32_t* rects_h, int64_t A)
{
    if (!ctx || !anchors_h || !rects_h || A <= 0) return pp_fail(ctx, PP_E_ARG, "pp_set_anchors: bad argument");
    PP_HIP(hipSetDevice(ctx->device));
    const int H = ctx->H, W = ctx->W;
    const int64_t HW = (int64_t)H * W;
    if (A % HW != 0 || A / HW != ctx->cfg.num_anchor_per_loc)
        return pp_fail(ctx, PP_E_ARG, "pp_set_anchors: A must be num_anchor_per_loc * (gx/2) * (gy/2)");
    for (int i = 0; i < 4 * A; ++i) {
        int lim = (i & 1) ? ctx->gy : ctx->gx;
        if (rects_h[i] < 0 || rects_h[i] >= lim) return pp_fail(ctx, PP_E_ARG, "pp_set_anchors: rectangle outside the grid");
    }
    for (void** q : {(void**)&ctx->anchors, (void**)&ctx->rects, (void**)&ctx->rect_x, (void**)&ctx->rect_y})
        if (*q) { (void)hipFree(*q); *q = nullptr; }
    PP_HIP(dalloc(&ctx->anchors, (size_t)A * 7));
    PP_HIP(hipMemcpy(ctx->anchors, anchors_h, (size_t)A * 7 * sizeof(float), hipMemcpyHostToDevice));
    PP_HIP(dalloc(&ctx->rects, (size_t)A * 4));
    PP_HIP(hipMemcpy(ctx->rects, rects_h, (size_t)A * 4 * sizeof(int32_t), hipMemcpyHostToDevice));
    // separable form: (minx,maxx) depends on (type, ix) only, (miny,maxy) on (type, iy) only
    const int types = (int)(A / HW);
    std::vector<int32_t> rx((size_t)types * H * 2), ry((size_t)types * W * 2);
    bool sep = true;
    for (int t = 0; t < types && sep; ++t)
        for (int ix = 0; ix < H && sep; ++ix)
            for (int iy = 0; iy < W; ++iy) {
                const int32_t* r = rects_h + 4 * (((int64_t)t * H + ix) * W + iy);
                if (iy == 0) { rx[((size_t)t * H + ix) * 2] = r[0]; rx[((size_t)t * H + ix) * 2 + 1] = r[2]; }
                if (ix == 0) { ry[((size_t)t * W + iy) * 2] = r[1]; ry[((size_t)t * W + iy) * 2 + 1] = r[3]; }
                if (r[0] != rx[((size_t)t * H + ix) * 2] || r[2] != rx[((size_t)t * H + ix) * 2 + 1] ||
                    r[1] != ry[((size_t)t * W + iy) * 2] || r[3] != ry[((size_t)t * W + iy) * 2 + 1]) { sep = false; break; }
            }
    ctx->rect_separable = sep ? 1 : 0;
    if (sep) {
        PP_HIP(dalloc(&ctx->rect_x, rx.size()));
        PP_HIP(dalloc(&ctx->rect_y, ry.size()));
        PP_HIP(hipMemcpy(ctx->rect_x, rx.data(), rx.size() * sizeof(int32_t), hipMemcpyHostToDevice));
        PP_HIP(hipMemcpy(ctx->rect_y, ry.data(), ry.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    }
    ctx->A = A;
    return 0;
}
