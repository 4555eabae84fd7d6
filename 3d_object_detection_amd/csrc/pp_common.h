// Internal definitions shared by the HIP sources of libpp_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <map>
#include <string>
#include <vector>
#include "../../include/pp_hip.h"

#define PP_EMPTY 0x7F7F7F7F  // "no index" sentinel; hipMemsetAsync(…, 0x7F, …) produces it

#define PP_HIP(call)                                                         \
    do {                                                                     \
        hipError_t e_ = (call);                                              \
        if (e_ != hipSuccess) return pp_fail_hip(ctx, e_, #call, __FILE__, __LINE__); \
    } while (0)

// ---- frame tables of the batched integer stages (pp_infer_batch) ---------------------------------
// The voxeliser / mask / PFN / post-processing kernels of a batch run as ONE launch per stage with
// blockIdx.z = frame; each frame's buffers are looked up in a device-resident table built once.
// Device side: the table entries are typed as GLOBAL pointers.  A pointer loaded from memory carries no address space, so every
// access through these tables used to be flat_load / flat_store (both wait counters, conservative s_waitcnt 0 around each); typed,
// the inlined stage bodies get global_load / global_store.  Same 8-byte layout on the host, which fills the tables.
#if defined(__HIP_DEVICE_COMPILE__)
#define PP_GP __attribute__((address_space(1)))
#else
#define PP_GP
#endif
#define PP_SET(dst, src) dst = (decltype(dst))(src) // host-side fill (the host functions are parsed in the device pass as well)
struct pp_pre_frame {
    int32_t PP_GP *pt_cell, *cell_first, *wave_cnt, *pt_rank, *slots, *scalars, *occ;
    float PP_GP* voxels;
    int32_t PP_GP *coors, *npts, *num;
    uint8_t PP_GP* mask;
    float PP_GP* feat;
    int32_t PP_GP* pmap;
};
struct pp_post_frame {
    const float PP_GP *cls, *box, *dir;
    const uint8_t PP_GP* mask;
    uint64_t PP_GP *cand, *shortl, *sel, *nmask;
    int32_t PP_GP *counters, *hist, *dirl;
    float PP_GP *boxes, *nbox;
};
#define PP_GROUP 32 // frames per batched launch of the integer stages (kernel-argument table size: 384 B of the 4 KB)
struct pp_in_group {
    const float* pts[PP_GROUP];
    int32_t n[PP_GROUP];
};

struct pp_tensor_h {
    std::vector<int64_t> shape;
    std::vector<float> data;
};

// one conv-like layer after packing (see conv.hip)
struct pp_layer {
    float* w = nullptr;        // packed weights on device
    int cin = 0, cout = 0;     // logical channels (cout = virtual channels for deconv)
    int ks = 1, stride = 1;    // conv kernel / stride (deconv: ks = 1, up = upsample factor)
    int up = 1;
    float* bn_scale = nullptr; // BatchNorm variant: per-input-channel affine folded for the prologue
    float* bn_shift = nullptr;
};

struct pp_slot {
    int32_t* cell_first = nullptr;  // [gx*gy*gz] first point index per cell
    int32_t* pt_cell = nullptr;     // [max_points]
    int32_t* pt_rank = nullptr;     // [max_points] pillar rank of first points
    int32_t* wave_cnt = nullptr;    // [max_points/64 + 8]
    int32_t* slots = nullptr;       // [max_voxels*T] ordered point indices per pillar
    int32_t* vox_scalars = nullptr; // [4]: istar, P_all
    int32_t* occ = nullptr;         // [gx*gy] occupancy -> summed-area table
    void* post = nullptr;           // pp_post workspace (postprocess.hip)
};

struct pp_ctx {
    pp_config cfg;
    int device = 0;
    std::string err;
    // ---- geometry ----
    int gx = 0, gy = 0, H = 0, W = 0; // BEV grid and level-1 feature map (H = gx/2 along x, W = gy/2 along y)
    int max_batch = 1;                // frames per batched launch (cfg.max_batch)
    int64_t A = 0;                    // anchors
    // ---- per-frame scratch of the integer stages: one slot per frame of a batch (the batched stage kernels look a frame's
    //      buffers up in device tables built from these) ----
    std::vector<pp_slot> slot;     // [max_batch]; the single-stage entry points use slot 0
    pp_pre_frame* d_pre = nullptr;   // [max_batch] device tables (built by pp_build_tables for the current anchor count)
    pp_post_frame* d_post = nullptr;
    int64_t tab_A = -1;
    float* anchors = nullptr;      // [A,7]
    int32_t* rect_x = nullptr;     // separable cell rectangles: [types, H, 2] (minx,maxx) / [types, W, 2] (miny,maxy)
    int32_t* rect_y = nullptr;
    int32_t* rects = nullptr;      // full [A,4] table (fallback when not separable)
    int rect_separable = 0;
    // ---- frame buffers (pp_infer_frame) ----
    float* f_voxels = nullptr; int32_t* f_coors = nullptr; int32_t* f_npts = nullptr; int32_t* f_num = nullptr;
    float* f_feat = nullptr; float* f_canvas = nullptr; uint8_t* f_mask = nullptr; int32_t* f_pmap = nullptr;
    float* f_cls = nullptr; float* f_box = nullptr; float* f_dir = nullptr;
    // ---- network ----
    std::map<std::string, pp_tensor_h> host_w;
    bool weights_ready = false;
    float* pfn_w = nullptr;     // [9][64] transposed
    float* pfn_scale = nullptr; // [64]
    float* pfn_shift = nullptr; // [64]
    void* net = nullptr;        // opaque pp_net (conv.hip)
    int precision = 0;          // pp_set_precision: 0 fp32 MFMA, 1 split-bf16 (bf16x3), 2 bf16, 3 fp16 operands -- convs, upsamplers and head
    // ---- measurement (pp_profile_begin/end) ----
    bool prof_on = false;
    std::vector<hipEvent_t> prof_ev; // start/stop pairs
    size_t prof_used = 0;
    double prof_flops = 0.0;
    // ---- per-stage timing (pp_stage_profile_begin/end): one event per stage boundary on the caller's stream ----
    bool stage_on = false;
    std::vector<hipEvent_t> stage_ev;
    std::vector<int> stage_id; // stage that FOLLOWS event i (-1: end of the pass)
    std::vector<hipStream_t> stage_stream; // stream event i was recorded on (an interval needs both ends on one stream)
    size_t stage_used = 0;
};
// stage ids of pp_stage_mark / pp_stage_profile_end
enum { PP_ST_VOXELIZE = 0, PP_ST_MASK = 1, PP_ST_PFN = 2, PP_ST_CONV = 3, PP_ST_NORM = 4, PP_ST_HEAD = 5, PP_ST_POST = 6 /* filter + threshold + gather */,
       PP_ST_POST_TOPK = 7 /* exact top-k + decode */, PP_ST_POST_NMS = 8 /* mask + greedy sweep + flip / range / compaction */, PP_ST_COUNT = 12 };
int pp_stage_mark(pp_ctx* ctx, hipStream_t stream, int id); // no-op unless stage profiling is on

int pp_fail_hip(pp_ctx* ctx, hipError_t e, const char* what, const char* file, int line);
int pp_fail(pp_ctx* ctx, int code, const char* msg);

// stage entry points implemented per file (all enqueue on stream, no sync)
int pp_net_create(pp_ctx* ctx);
void pp_net_destroy(pp_ctx* ctx);
int pp_net_commit(pp_ctx* ctx);
int pp_post_create(pp_ctx* ctx);
// slot-aware internals of the public single-frame entry points
int pp_voxelize_slot(pp_ctx* ctx, int s, const float* pts, int n, int nfeat, float* voxels, int32_t* coors, int32_t* npts,
                     int32_t* num_pillars, hipStream_t stream);
int pp_anchor_mask_slot(pp_ctx* ctx, int s, const int32_t* coors, const int32_t* num_pillars, uint8_t* mask, hipStream_t stream);
int pp_postprocess_slot(pp_ctx* ctx, int s, const float* cls, const float* box, const float* dir, const uint8_t* mask, float* det,
                        int32_t* det_count, int nms_mode, hipStream_t stream);
// nb canvases (or, when pmap != nullptr, nb sparse BEV inputs: pillar-index maps + PFN rows) -> pre-norm [nb,320,H,W] + stats
int pp_run_backbone(pp_ctx* ctx, const float* canvas, int nb, hipStream_t stream, const int32_t* pmap, const float* feat);
int pp_pillar_map(pp_ctx* ctx, const int32_t* coors, const int32_t* num_pillars, int32_t* pmap, hipStream_t stream);
// batched integer stages: frames b0 .. b0+g-1 of a batch, one launch per stage (blockIdx.z = frame)
int pp_build_tables(pp_ctx* ctx);
void pp_post_fill_table(pp_ctx* ctx, int slot, pp_post_frame* f);
int pp_voxelize_group(pp_ctx* ctx, int b0, int g, const pp_in_group& in, hipStream_t stream);
int pp_anchor_mask_group(pp_ctx* ctx, int b0, int g, hipStream_t stream);
int pp_pfn_pmap_group(pp_ctx* ctx, int b0, int g, hipStream_t stream);
int pp_postprocess_group(pp_ctx* ctx, int b0, int g, float* det, int32_t* det_count, int nms_mode, hipStream_t stream);
int pp_run_head_fused(pp_ctx* ctx, float* cls, float* box, float* dir, int nb, hipStream_t stream); // norm+ReLU fused in the prologue
void pp_post_destroy(pp_ctx* ctx);

static inline int pp_div_up(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }
