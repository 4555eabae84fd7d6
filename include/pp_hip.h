/* pp_hip.h -- C ABI of libpp_hip.so: the MI355X (gfx950) PointPillars inference hot path.
 *
 * The reference (1005088h/3d_object_detection) is pure Python and has no FFI; the
 * boundary it offers is the Python call surface of train.py:192-196,224-230.  Each entry
 * point below is what a ctypes binding for that surface binds (INTEGRATION.md shows the
 * stubs); the "replaces" note cites the reference function it stands in for.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer unless its name ends in _h (host);
 *  - the caller allocates every output; functions only enqueue work on `stream`
 *    (a hipStream_t passed as void*; NULL = the default stream) and never synchronise,
 *    except where noted;
 *  - return value: 0 = ok, <0 = -(hipError_t), >0 = argument/state error (PP_E_*);
 *    pp_last_error(ctx) returns a static, human readable message for the last failure;
 *  - one pp_ctx per GPU/stream; a ctx is not re-entrant, different ctxs are independent.
 */
#ifndef PP_HIP_H
#define PP_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PP_MAX_CLASSES 12 /* class ranges of the anchor table (reference: 3; the build-side nuScenes table: 10) */
#define PP_E_ARG 1      /* bad argument (null pointer, size out of range) */
#define PP_E_STATE 2    /* weights / anchors not loaded yet */
#define PP_E_NAME 3     /* unknown weight name or wrong shape */

typedef struct pp_ctx pp_ctx;

/* Geometry and limits.  Filled by the Python VoxelGenerator/AnchorAssigner mirrors from
 * the same config keys the reference reads (voxel_generator.py:6-26, inference.py:13-19). */
typedef struct pp_config {
    float voxel_size[3];
    float offset[3];          /* detection_offset */
    int32_t grid_size[3];     /* gx, gy, gz (gz must be 1 for the BEV path) */
    int32_t max_voxels;
    int32_t max_num_points;   /* T */
    int32_t num_point_features; /* F: must be 4 for the PFN (its 9 input features x, y, z, i, 3 offsets from the pillar mean, 2 from the
                                 * pillar centre -- pointpillars8_shared.py:30-60 -- and the (64, 9, 1) weight of the state_dict fix it:
                                 * pp_pfn / pp_infer_* refuse other values); pp_voxelize alone takes any F >= 3 */
    int32_t max_points;       /* capacity of the per-point workspace (N upper bound) */
    int32_t num_anchor_per_loc; /* anchors per BEV location = head geometry: cls na, box 7*na, dir 2*na rows (reference: 9) */
    int32_t num_classes;      /* <= PP_MAX_CLASSES (reference: 3) */
    int32_t class_begin[PP_MAX_CLASSES]; /* anchor index ranges, class_masks of anchor_assigner.py:289 */
    int32_t class_end[PP_MAX_CLASSES];
    double center_limit[6];   /* compared in fp64 like inference.py:105-109 (python floats) */
    int32_t norm_kind;        /* 0 = InstanceNorm2d(eps 1e-3) backbone (_shared), 1 = BatchNorm2d (_export/_trt) */
    int32_t nms_pre_max;      /* 1000 */
    int32_t nms_post_max;     /* 300 (<= 1024) */
    float nms_iou_threshold;  /* 0.1  */
    float score_threshold;    /* 0.05 */
    int32_t max_batch;        /* frames per batched launch of pp_infer_batch (>= 1; 0 is read as 1) */
} pp_config;

/* Lifetime.  pp_create allocates all device workspace for the configured sizes. */
pp_ctx* pp_create(int device, const pp_config* cfg);
void pp_destroy(pp_ctx* ctx);
const char* pp_last_error(pp_ctx* ctx); /* ctx may be NULL: last pp_create failure */

/* Weights by state_dict key (replaces net.load_state_dict, train.py:201-202; key names of
 * networks/pointpillars8_shared.py:346-357).  host_ptr is fp32, contiguous.  Call
 * pp_commit_weights once after the last tensor: it folds BN, repacks for the kernels
 * and uploads.  Synchronous. */
int pp_load_weights(pp_ctx* ctx, const char* name, const void* host_ptr_h, const int64_t* shape_h, int ndim);
int pp_commit_weights(pp_ctx* ctx);

/* MFMA operand type of the network behind the PFN -- every 3x3 convolution (stride 1 and 2), the three ConvTranspose(k = s)
 * upsamplers and the head -- SURVEY 8(f).4; the reference's deployed path is TensorRT FP16 (framework/trt_utils.py:30,
 * networks/pointpillars8_trt.py:208-223,295-314):
 *   0 fp32 MFMA (default, exact, the parity path);
 *   1 split-bf16 "bf16x3" (x = hi + lo, three bf16 MFMAs per product, fp32 accumulation: fp32-equivalent for this network,
 *     meets the fp32 parity bar -- DESIGN.md);
 *   2 bf16 operands;   3 fp16 operands (the reference's deploy arithmetic) -- own tolerance table in DESIGN.md;
 *   4 "fp16s": mode 3 plus fp16 STORAGE of every activation tensor behind the first convolution -- the level buffers (block-head
 *     outputs, Resnet2 intermediates, residuals) and the [320,H,W] concat buffer between the upsamplers and the head; TensorRT FP16
 *     engines keep fp16 tensors between layers.  The first conv still reads the fp32 PFN rows / canvas, the logits stay fp32, and the
 *     InstanceNorm statistics are accumulated from the UNROUNDED fp32 values of a layer while its consumer normalises the fp16-rounded
 *     tensor (DESIGN.md tolerance table).  Needs maps that are multiples of 4 pixels wide at all three levels (W a multiple of 16,
 *     H * W of 64) and the 9-anchor head; otherwise the context runs mode 3 -- pp_effective_precision tells which.
 * Modes 1 - 3: accumulation is fp32 and every activation stays fp32 NCHW in HBM (normalise + ReLU in fp32, round while staging).
 * A layer whose shape none of the 16-bit tilings takes (maps not a multiple of 4 wide, Cin not a multiple of 16 / 32) keeps its
 * fp32 tiling: pp_layer_tilings reports what runs.  Call before pp_commit_weights (a change of mode invalidates the committed
 * weights until the next commit). */
int pp_set_precision(pp_ctx* ctx, int mode);
/* The mode the committed launch plan really runs (pp_set_precision's value, except 4 -> 3 where the fp16 tensors are not possible);
 * valid after pp_commit_weights, -1 before. */
int pp_effective_precision(pp_ctx* ctx);

/* Anchor table built by the host mirror of AnchorAssigner.__init__ (anchor_assigner.py:221-298):
 * anchors f32[A,7], cell rectangles i32[A,4] from get_anchor_coor (box_np_ops.py:288-305). Synchronous. */
int pp_set_anchors(pp_ctx* ctx, const float* anchors_h, const int32_t* anchor_rects_h, int64_t num_anchors);

/* replaces VoxelGenerator.generate / points_to_voxels (voxel_generator.py:28-40,82-106).
 * pts f32[n,nfeat] -> voxels f32[max_voxels,T,F] (rows >= *num_pillars untouched), coors i32[max_voxels,3]
 * (x,y,z), npts i32[max_voxels], num_pillars i32[1]. Bit-exact incl. the max_voxels break. */
int pp_voxelize(pp_ctx* ctx, const float* pts, int n, int nfeat, float* voxels, int32_t* coors,
                int32_t* npts, int32_t* num_pillars, void* stream);

/* replaces AnchorAssigner.create_mask (anchor_assigner.py:322-335; box_np_ops.py:168-257).
 * mask u8[A] (0/1). */
int pp_anchor_mask(pp_ctx* ctx, const int32_t* coors, const int32_t* num_pillars, uint8_t* mask, void* stream);

/* replaces PointNet.forward (pointpillars8_shared.py:30-60). feat f32[max_voxels,64], rows < *num_pillars written. */
int pp_pfn(pp_ctx* ctx, const float* voxels, const int32_t* coors, const int32_t* npts,
           const int32_t* num_pillars, float* feat, void* stream);

/* replaces PointPillarsScatter.forward (pointpillars8_shared.py:76-111; CUDA scatter of
 * pointpillars8_trt.py:176-193). canvas f32[64,gx,gy], fully written (zero fill + scatter). */
int pp_scatter(pp_ctx* ctx, const float* feat, const int32_t* coors, const int32_t* num_pillars,
               float* canvas, void* stream);

/* replaces RPN.forward (pointpillars8_shared.py:173-181). canvas f32[64,gx,gy] -> rpn_out f32[320,gx/2,gy/2]. */
int pp_backbone(pp_ctx* ctx, const float* canvas, float* rpn_out, void* stream);

/* replaces SharedHead.forward (pointpillars8_shared.py:323-343). rpn_out f32[320,H,W] ->
 * cls f32[A] , box f32[A,7], dir f32[A,2], ordered (anchor type, x, y). */
int pp_head(pp_ctx* ctx, const float* rpn_out, float* cls, float* box, float* dir, void* stream);

/* replaces Inference.infer_gpu (inference.py:26-138). det f32[num_classes*nms_post_max, 9] rows
 * (x,y,z,l,w,h,r,score,class) grouped by class in class order, det_count i32[1+num_classes]
 * (total, then per class). nms_mode 0 = axis-aligned "+1" NMS (nms.py), 1 = rotated (eval/iou.py). */
int pp_postprocess(pp_ctx* ctx, const float* cls, const float* box, const float* dir, const uint8_t* mask,
                   float* det, int32_t* det_count, int nms_mode, void* stream);

/* First half of pp_postprocess as a stage of its own, behind Inference.infer_torch (inference.py:140-189: per class
 * mask gather, sigmoid, score >= 0.05, top-nms_pre_max).  idx i32[num_classes][nms_pre_max] anchor ids by descending score
 * (ties: lower anchor id; -1 beyond a class's count), score f32[num_classes][nms_pre_max], count i32[num_classes]. */
int pp_select_candidates(pp_ctx* ctx, const float* cls, const float* box, const float* dir, const uint8_t* mask,
                         int32_t* idx, float* score, int32_t* count, void* stream);

/* Fused frame: voxelise -> mask -> PFN -> BEV -> backbone -> head -> post-process, no host sync. */
int pp_infer_frame(pp_ctx* ctx, const float* pts, int n, float* det, int32_t* det_count, int nms_mode, void* stream);

/* nb <= cfg.max_batch independent frames in ONE pass: the conv/deconv/head launches carry the frame as grid.z
 * (per-frame InstanceNorm statistics; same detections as nb calls of pp_infer_frame, logits equal up to the
 * fp32 summation order of those statistics, ~3e-5), the small integer
 * stages run per frame on the same stream.  pts_h / n_h: HOST arrays of nb device pointers / point counts.
 * det f32[nb][num_classes*nms_post_max][9], det_count i32[nb][PP_DET_COUNT_STRIDE] (total, then per class). */
#define PP_DET_COUNT_STRIDE (1 + PP_MAX_CLASSES)
int pp_infer_batch(pp_ctx* ctx, const float* const* pts_h, const int32_t* n_h, int nb, float* det, int32_t* det_count,
                   int nms_mode, void* stream);

/* Inspection hook for the parity tests: copies one tensor of frame `frame` of the LAST pp_infer_batch / pp_infer_frame
 * pass out of the context's internal buffers into caller memory (device pointer, enqueued on stream).
 * kind 0 cls f32[A] | 1 box f32[A,7] | 2 dir f32[A,2] | 3 anchor mask u8[A] | 4 rpn output f32[320,H,W] as RPN.forward
 * returns it (pointpillars8_shared.py:173-181; the fused path never stores it, it is materialised for the copy) |
 * 5 PFN rows f32[max_voxels,64] | 6 coors i32[max_voxels,3] | 7 pillar count i32[1]. */
int pp_fetch_frame_tensor(pp_ctx* ctx, int frame, int kind, void* dst, void* stream);

/* Stateless box ops (replace framework/box_torch_ops.py:18-77 and framework/nms.py:6-40,
 * eval/iou.py:438-473). */
int pp_box_decode(const float* enc, const float* anchors, float* out, int64_t n, void* stream);
int pp_corners2d(const float* centers, const float* dims, const float* angles /* may be NULL */, float* corners /*[n,4,2]*/,
                 int64_t n, void* stream);
int pp_standup2d(const float* corners /*[n,4,2]*/, float* boxes /*[n,4]*/, int64_t n, void* stream);
/* dets f32[n,stride] (stride 5: x1,y1,x2,y2,score; 6: cx,cy,dx,dy,angle,score); keep i32[n], nkeep i32[1].
 * Sorts by score (desc, ties by lower index), builds the 64x64 bitmask tiles, greedy sweep on device. */
int pp_nms(const float* dets, int n, int stride, float thresh, int32_t* keep, int32_t* nkeep, int rotate, void* stream);
int pp_rotated_iou(const float* boxes_a /*[n,5]*/, const float* boxes_b /*[m,5]*/, float* iou /*[n,m]*/, int n, int m, void* stream);

/* ---- ROS ingest ----
 * pp_unpack_points replaces `np.asarray(list(pc2.read_points(msg)))[:, :4].astype(np.float32)` (ros_node.py:55-59):
 * the first four fields of a sensor_msgs/PointCloud2 payload -> f32[n,4] on the device.  `data` is the message's byte
 * buffer in device memory; point i sits at (i / width) * row_step + (i % width) * point_step; offs/dtypes are HOST
 * arrays of the four fields' byte offsets and PointField datatype codes (1 INT8, 2 UINT8, 3 INT16, 4 UINT16,
 * 5 INT32, 6 UINT32, 7 FLOAT32, 8 FLOAT64). */
int pp_unpack_points(const void* data, int64_t n, int64_t width, int64_t row_step, int point_step, const int32_t* offs /*[4]*/,
                     const int32_t* dtypes /*[4]*/, int big_endian, float* out /*[n,4]*/, void* stream);

/* ---- evaluation ----
 * pp_rotated_iou_eval replaces rotate_iou_gpu_eval (eval/iou.py:540-638): out[i,j] for box i and query j with
 * criterion -1: IoU, 0: inter / area(query), 1: inter / area(box), 2: intersection area -- the reference's kernel
 * hands the QUERY box to devRotateIoUEval first (:600-603).  Device pointers, boxes are (cx,cy,dx,dy,angle).
 * pp_eval_statistics / pp_eval_fused_statistics replace the numba host loops compute_statistics_jit and
 * fused_compute_statistics (eval/eval.py:62-119,182-216); HOST pointers, no GPU work.  overlaps is row-major
 * [det rows][gt columns] with `ov_ld` doubles per row; ignored_* are the reference's -1/0/1 codes.
 * thresholds_out must hold gt_size doubles; pr is [nthresh][4] and accumulated into (tp, fp, fn, unused). */
int pp_rotated_iou_eval(const float* boxes /*[n,5]*/, const float* qboxes /*[k,5]*/, float* out /*[n,k]*/, int n, int k, int criterion,
                        void* stream);
int pp_eval_statistics(const double* overlaps, int64_t ov_ld, int det_size, int gt_size, const int64_t* ignored_gt,
                       const int64_t* ignored_det, const float* dt_scores, double min_overlap, double thresh, int compute_fp,
                       int64_t* tp_fp_fn /*[3]*/, double* thresholds_out, int64_t* n_thresholds);
int pp_eval_fused_statistics(const double* overlaps, int64_t ov_ld, double* pr, const int64_t* gt_nums, const int64_t* dt_nums,
                             int n_frames, const int64_t* ignored_gts, const int64_t* ignored_dets, const float* dt_scores,
                             double min_overlap, const double* thresholds, int n_thresholds);

/* One (class, overlap threshold) cell of the AP table in one call: the loop nest of eval/eval.py:396-434 (per-frame
 * compute_statistics_jit at threshold 0, get_thresholds :42-59, fused_compute_statistics per part, precision / recall and
 * the running maximum).  parts_h[j]: row-major [detections of part j][ground truths of part j] overlaps; part_frames_h:
 * frames per part; *_nums_h per frame; ignored_* (-1/0/1 codes) and dt_scores concatenated in frame order.
 * precision_h / recall_h: n_sample_pts (41) doubles, zero behind the last threshold like the reference's np.zeros. */
int pp_eval_class_ap(const double* const* parts_h, const int64_t* part_frames_h, int n_parts, const int64_t* dt_nums_h,
                     const int64_t* gt_nums_h, int64_t n_frames, const int64_t* ignored_gt_h, const int64_t* ignored_dt_h,
                     const float* dt_scores_h, double min_overlap, int64_t num_valid_gt, int n_sample_pts, double* precision_h,
                     double* recall_h);

/* Measurement hooks for bench.py: between begin and end every launch of the dominant kernel
 * (conv3x3 stride 1 on the level-0 map, 3 launches per frame) is bracketed by hipEvents on the
 * launch stream.  pp_profile_end synchronises the events and reports the average duration (ms),
 * the number of launches seen and the algorithmic FLOPs of one launch (2*H*W*Cin*Cout*9). */
int pp_profile_begin(pp_ctx* ctx);
int pp_profile_end(pp_ctx* ctx, double* avg_ms_h, int32_t* launches_h, double* flops_per_launch_h);
/* executed MFMA flops / algorithmic (direct-convolution) flops of that layer's tiling (Winograd F(2x2,3x3): 4/9) */
double pp_dominant_executed_ratio(pp_ctx* ctx);
/* Per-stage GPU time of the fused path: between begin and end every pp_infer_batch pass records one event per stage
 * boundary on its stream (and so does a stand-alone pp_postprocess); end synchronises and sums the milliseconds
 * per stage into ms_h[12]: 0 voxelise, 1 anchor mask, 2 PFN + pillar map, 3 conv / deconv launches (+ statistics
 * finalisation), 4 norm_relu_stats, 5 head, 6 post-processing filter (mask, sigmoid, threshold, candidate gather),
 * 7 exact top-k + box decode, 8 NMS + direction flip / range mask / compaction (9-11 unused).
 * Meant for an untimed side pass of bench.py and for the drop-in classes' p1..p4 / stage timers. */
int pp_stage_profile_begin(pp_ctx* ctx);
int pp_stage_profile_end(pp_ctx* ctx, double* ms_h);
/* The network's launch plan as text, one line per conv / deconv / head layer in execution order:
 * "<index> kind=<0 conv3x3|1 deconv|2 head> cin= cout= stride= up= level= wino=<0 direct|1,2,4 Winograd|3 1x1 GEMM> tiling=<name>".
 * Returns the text length (buf may be NULL to query). */
int pp_layer_tilings(pp_ctx* ctx, char* buf_h, int cap);
/* The autotuner's table (process-wide) as text, "layer signature<TAB>tiling" per line.  Rank 0 of a multi-GPU job
 * tunes, exports and broadcasts it; the other ranks import it BEFORE pp_commit_weights so all ranks run identical
 * kernels.  pp_tune_export returns the text length (buf may be NULL to query), pp_tune_import the lines taken. */
int pp_tune_export(char* buf_h, int cap);
int pp_tune_import(const char* text_h);
/* name of the tiling the autotuner chose for that dominant layer (static string owned by ctx) */
const char* pp_dominant_kernel(pp_ctx* ctx);
int pp_version(void);

#ifdef __cplusplus
}
#endif
#endif /* PP_HIP_H */
