#!/usr/bin/env python3
"""bench.py -- frames/s of the PointPillars hot path (voxelise -> NMS) on MI355X.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A step = one pass of `--batch` independent frames per GPU through pp_infer_batch (each frame keeps the
reference's batch=1 semantics, train.py:219-242: no statistic is shared between frames; the frames only share
kernel launches) plus the async D2H of their detection records.  Workload: configs/<config>.json, synthetic
LiDAR-shaped clouds, random-init weights of the reference architecture.

Two timed regions of K steps each, both bracketed by barrier + synchronize, max over ranks:
  * `value`            -- clouds already resident in HBM when the region starts (the contract's definition of `value`);
  * `value_host_start` -- SURVEY 8(d)'s timer, the one train.py:223-237 uses: clouds start in pinned HOST memory, their
                          H2D copies run inside the region (copy stream, double-buffered, overlapping the previous pass),
                          the region ends when the last detection record is on the host.
Frames are sharded by index across ranks with no data-path collective; RCCL only gathers the detection records, one
all_gather per step inside the timed loop (stream-ordered behind the step's kernels).  Default: weak scaling, `--batch`
frames per rank per step.
`--global-batch G` (BASELINE config 5: G = 64) fixes the TOTAL frames per step and shards them, strong scaling.
Rank 0 prints ONE JSON line with `roofline` (dominant conv kernel timed with HIP events on its launch stream inside
the resident region; `frac` = EXECUTED MFMA flops / fp32-MFMA peak; per-stage HBM rooflines under `stages`) and, at
N=1, `cpu_baseline` (the CPU oracle on the host cores) and `extras` (batch-1 latency, trained-like head bias).
"""
import argparse
import glob
import hashlib
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np
import torch

F32_MFMA_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32 dense peak
LP_MFMA_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: bf16 / fp16 MFMA dense peak (reduced-precision deploy modes, tagged lines only)
HBM_PEAK_GBS = 8000.0         # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.3 TB/s achievable)
CLOUD_DESC = {"eight_20cm": "KITTI-shape 64-beam 20k-point", "ntusl_10cm": "64-beam two-sweep 60k-point", "nuscene": "nuScenes-shape 32-beam 34k-point"}


def source_hash():
    """sha256 over the HIP sources the library was built from: ties a committed PMC figure to the build it was taken on."""
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, "3d_object_detection_amd", "csrc", "*.hip")) + glob.glob(os.path.join(ROOT, "3d_object_detection_amd", "csrc", "*.h"))):
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def hbm_traffic(tiling, frames, H, W):
    """HBM bytes per launch of the roofline kernel.  PMC counters cannot be read from inside this process, so the figure
    comes from the committed rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this same command (tools/profile_round.sh +
    tools/hbm_traffic.py; FETCH_SIZE doubled per the gfx950 correction of MI355X_MICROARCH.md) -- and only if they were
    taken for the layer shape, the tiling, the batch AND the source tree that is running; otherwise null."""
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_hbm_traffic.json")), reverse=True):
        try:
            with open(path) as f:
                t = json.load(f)
            e = t["tilings"].get(tiling)
            if e is None or t["frames_per_launch"] != frames or t.get("source_hash") != source_hash() or not t.get("layer", "").endswith(f"@ {H}x{W}"):
                continue
            return int(e["hbm_bytes_per_launch"]), os.path.relpath(path, ROOT) + " (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, bytes/launch)"
        except (OSError, KeyError, ValueError):
            continue
    return None, None


def host_cores():
    """CPU threads this process may really use: affinity, capped by the cgroup CPU quota (a GPU box
    gives one GPU's share of the host, 16 CPUs, although 256 are visible)."""
    n = os.cpu_count() or 1
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        pass
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(float(txt[0]) / float(txt[1]) + 0.5)))
            else:
                q = int(txt[0])
                if q > 0:
                    per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                    n = min(n, max(1, int(q / per + 0.5)))
            break
        except Exception:
            continue
    return min(n, int(os.environ.get("PP_CPU_THREADS", "16")))


def cpu_baseline(synth, config, n_frames=8):
    """The CPU oracle (oracle/: C for voxelise/mask/NMS, torch-CPU fp32 for PFN/backbone/head) on the
    host cores, bounded sample (1 warm + n_frames timed frames of the same workload)."""
    from oracle import c_oracle as C
    from oracle import pp_oracle as O
    cores = host_cores()
    torch.set_num_threads(cores)
    cfg = synth.load_config(config)
    s = O.voxel_setup(cfg)
    a = O.make_anchors(s)
    sd = synth.seeded_state_dict(0)
    C.lib()
    clouds = [synth.lidar_cloud(config, seed=1000 + i) for i in range(n_frames + 1)]

    def frame(pts):
        v, c, n = C.points_to_voxels(pts, s["voxel_size"], s["offset"], s["grid_size"], cfg["max_voxels"], cfg["max_num_points"])
        mask = C.create_mask(c, s["grid_size"], a["anchors_coors"])
        rpn = O.backbone(O.scatter(O.pfn(v, n, c, sd, s), c, s["grid_size"]), sd)
        cls, box, dr = O.head(rpn, sd)
        return O.postprocess(cls, box, dr, mask, a["anchors"], a["class_masks"], cfg["center_limit"], nms_fn=C.nms_aabb)

    frame(clouds[0])
    t0 = time.time()
    for p in clouds[1:]:
        frame(p)
    dt = time.time() - t0
    return {"value": round(n_frames / dt, 4), "unit": "frames/s", "cores": cores, "kind": "port",
            "sample": f"{n_frames} frames of {config} ({clouds[0].shape[0]}-pt clouds, batch=1) after 1 warm-up, oracle/ CPU path, {dt:.1f} s"}


def executed_factor(L):
    """Executed / algorithmic MFMA flops of a layer's tiling: Winograd F(2x2,3x3) executes 4/9; a split-bf16 ("bf16x3", p1)
    tiling issues three bf16 MFMAs per product."""
    if L["wino"] in (1, 2, 4):
        return 4.0 / 9.0
    if L["wino"] == 6:  # Winograd F(4x4,3x3): 36 multiplies per 4x4 tile against 144
        return 0.25
    if " p1" in L["tiling"]:
        return 3.0
    return 1.0


def frame_flops(H, W, tilings):
    """Algorithmic (direct-convolution) and executed MFMA flops of one frame from the launch plan (SURVEY 8(d) formula:
    203.2 GFLOP at eight_20cm)."""
    alg = ex = 0.0
    for L in tilings:
        hw = (H >> L["level"]) * (W >> L["level"])
        if L["kind"] == 0:
            f = 2.0 * hw * L["cin"] * L["cout"] * 9
        elif L["kind"] == 1:
            f = 2.0 * hw * L["cin"] * L["cout"] * L["up"] * L["up"]
        else:
            f = 2.0 * hw * L["cin"] * L["cout"]
        alg += f
        ex += f * executed_factor(L)
    return alg, ex


def stage_rooflines(eng, clouds, NB, cfg, passes=3, mfma_peak=F32_MFMA_PEAK_TFLOPS):
    """Untimed side pass with one HIP event per stage boundary (pp_stage_profile_*): GPU ms per stage and frame, against
    the algorithmic HBM bytes of SURVEY 8(d) (fp32) and the 8 TB/s peak; conv / head stages against the MFMA peak."""
    eng.infer_batch(clouds[:NB])
    torch.cuda.synchronize()
    P = float(np.mean([int(eng.fetch(b, "num").item()) for b in range(NB)]))
    N = float(np.mean([c.shape[0] for c in clouds[:NB]]))
    eng.stage_profile_begin()
    for _ in range(passes):
        eng.infer_batch(clouds[:NB])
    torch.cuda.synchronize()
    ms = eng.stage_profile_end()
    per_frame = {k: v / (passes * NB) for k, v in ms.items()}
    T = int(cfg["max_num_points"])
    gx, gy = int(eng.grid_size[0]), int(eng.grid_size[1])
    HW = eng.H * eng.W
    act_bytes = 2 if eng.effective_precision() == "fp16s" else 4  # the tensors norm_relu_stats reads and writes are fp16 under fp16s
    bytes_ = {
        "voxelize": 16 * N + P * (16 * T + 16),                       # points read + pillars (voxels, coors, count) written
        "anchor_mask": 3 * 4 * gx * gy + eng.A,                        # occupancy table written + two scan passes, mask written
        "pfn_pmap": P * 16 * T + 256 * P + 4 * gx * gy,                # pillars read, PFN rows + pillar map written
        "norm_relu_stats": 2 * act_bytes * HW * (64 + 128 / 4 + 256 / 16),  # block-head maps read + written once, 3 levels (stored element size)
        "postprocess": 4 * eng.A + eng.A + 3 * 1000 * (28 + 8 + 28),   # cls logits + mask read, box/dir/anchor rows of <= 1000 candidates per class
    }
    out = {}
    for k, b in bytes_.items():
        t = per_frame[k] * 1e-3
        gbs = b / t / 1e9 if t > 0 else 0.0
        assert gbs <= HBM_PEAK_GBS, (k, gbs, "a stage cannot move its algorithmic bytes faster than the HBM peak: the byte count is wrong")
        out[k] = {"bound": "hbm", "ms_per_frame": round(per_frame[k], 5), "algorithmic_bytes_per_frame": int(b), "achieved": round(gbs, 2),
                  "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 5)}
    til = eng.layer_tilings()
    alg, ex = frame_flops(eng.H, eng.W, til)
    head = [L for L in til if L["kind"] == 2]
    alg_h, _ = frame_flops(eng.H, eng.W, head)
    _, ex_h = frame_flops(eng.H, eng.W, head)
    for k, (a_, e_) in {"conv": (alg - alg_h, ex - ex_h), "head": (alg_h, ex_h)}.items():
        t = per_frame[k] * 1e-3
        out[k] = {"bound": "mfma", "ms_per_frame": round(per_frame[k], 5), "algorithmic_flops_per_frame": a_, "executed_flops_per_frame": e_,
                  "achieved": round(e_ / t / 1e12, 2) if t > 0 else 0.0, "peak": mfma_peak, "unit": "TFLOP/s",
                  "frac": round(e_ / t / 1e12 / mfma_peak, 4) if t > 0 else 0.0}
    parts = ("post_filter", "post_topk_decode", "post_nms")
    out["postprocess"]["parts_ms_per_frame"] = {k: round(per_frame[k], 5) for k in parts}
    total = sum(v for k, v in per_frame.items() if k not in parts)
    whole = {"gpu_ms_per_frame": round(total, 5), "algorithmic_gflop": round(alg / 1e9, 2), "executed_gflop": round(ex / 1e9, 2),
             "executed_tflops": round(ex / (total * 1e-3) / 1e12, 2) if total > 0 else 0.0,
             "executed_frac": round(ex / (total * 1e-3) / 1e12 / mfma_peak, 4) if total > 0 else 0.0,
             "algorithmic_tflops": round(alg / (total * 1e-3) / 1e12, 2) if total > 0 else 0.0,
             "mean_points": N, "mean_pillars": P}
    return out, whole, til


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--config", default="eight_20cm")
    ap.add_argument("--cls-bias", type=float, default=None, help="'trained-like' head bias (e.g. -4.6); default random init")
    ap.add_argument("--batch", type=int, default=32,
                    help="independent frames per pass per GPU (pp_infer_batch: frame = grid.z of the conv launches); weak scaling")
    ap.add_argument("--global-batch", type=int, default=0,
                    help="TOTAL frames per step, sharded by frame index over the ranks (strong scaling; BASELINE config 5: 64)")
    ap.add_argument("--precision", default="fp32", choices=["fp32", "bf16x3", "bf16", "fp16", "fp16s"],
                    help="MFMA operand type of convolutions, upsamplers and head (pp_set_precision); anything but fp32 prints a TAGGED line, never the headline")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-frames", type=int, default=20)
    ap.add_argument("--no-extras", action="store_true", help="skip the untimed side measurements (stage rooflines, batch-1 latency, trained-like bias)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # PP_BENCH_ENGINE=<module>: the CPU rehearsal of the rank flow (tests/test_bench_ranks.py) swaps the engine and the
    # device layer for a stub; never a measurement
    stub = importlib.import_module(os.environ["PP_BENCH_ENGINE"]) if os.environ.get("PP_BENCH_ENGINE") else None
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # PP_BENCH_BACKEND=gloo: rehearsal of the N-rank flow on a box with fewer GPUs than ranks (ranks share devices,
        # the gather goes through the host) -- RCCL refuses two ranks on one device.  Never a valid measurement.
        backend = "gloo" if stub else os.environ.get("PP_BENCH_BACKEND", "nccl")
        if not stub:
            if backend != "nccl":
                local = local % torch.cuda.device_count()
            torch.cuda.set_device(local)
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    dev = stub.device() if stub else torch.device("cuda", local)
    if not stub:
        torch.cuda.set_device(dev)
    D = stub if stub else importlib.import_module("3d_object_detection_amd.devlayer")

    synth = importlib.import_module("3d_object_detection_amd.synth")
    shard = importlib.import_module("3d_object_detection_amd.shard")
    eng_mod = stub if stub else importlib.import_module("3d_object_detection_amd.engine")
    cfg = synth.load_config(args.config)
    cfg["device"] = dev

    K, W = args.steps, args.warmup
    if args.global_batch > 0:
        mine = shard.frames_for_rank(rank, world, args.global_batch)  # this rank's frame indices of every step
        scaling = "strong"
    else:
        mine = [rank * args.batch + b for b in range(max(1, args.batch))]
        scaling = "weak"
    NB = len(mine)
    frames_per_step = args.global_batch if args.global_batch > 0 else world * NB
    MAXB = 64 if (args.batch > 32 and args.global_batch == 0) else 32  # frames per pp_infer_batch pass (pp_create takes up to 64: --batch 64 measured 1002 against 986 frames/s, at twice the memory and latency); a rank with more frames per step runs several passes
    passes = [mine[i:i + MAXB] for i in range(0, NB, MAXB)] if NB else []
    # every rank builds its engine for the SAME max_batch -- the busiest rank's frame count (a global batch the world size does
    # not divide gives the ranks different counts): the tuner's key and the Winograd strip decision carry it, so rank 0's
    # table fits every rank and all ranks run identical kernels
    NB_MAX = shard.frames_per_rank_max(world, args.global_batch) if args.global_batch > 0 else NB
    ENG_B = max(1, min(NB_MAX, MAXB))

    def make_engine():
        e = eng_mod.Engine(dict(cfg), device_index=local, max_batch=ENG_B, **({} if stub else {"precision": args.precision}))
        e.load_state_dict(synth.seeded_state_dict(0, cls_bias=args.cls_bias, num_anchor_per_loc=getattr(e, "num_anchor_per_loc", 9)))
        return e

    # rank 0 tunes (pp_commit_weights measures the tilings on the device), the others import its table first:
    # every rank runs identical kernels
    eng = make_engine() if rank == 0 else None
    if dist:
        shard.share_tuning(eng_mod.tuning_lib(), src=0)
    if eng is None:
        eng = make_engine()

    # one distinct cloud per frame of a step (a pool smaller than the pass re-reads cache-warm inputs)
    host = [D.pin(torch.from_numpy(synth.lidar_cloud(args.config, seed=1000 + f))) for f in mine]
    clouds = [h.to(dev) for h in host]
    stage_bufs = [[torch.empty_like(c) for c in clouds] for _ in range(2)]
    rows = eng.cfg.num_classes * eng.cfg.nms_post_max
    ncnt = eng.cnt_stride
    det = torch.zeros((K, max(NB, 1), rows, 9), dtype=torch.float32, device=dev)
    cnt = torch.zeros((K, max(NB, 1), ncnt), dtype=torch.int32, device=dev)
    det_h = D.pin(torch.zeros((K, max(NB, 1), rows, 9), dtype=torch.float32))
    cnt_h = D.pin(torch.zeros((K, max(NB, 1), ncnt), dtype=torch.int32))
    compute, copier = D.stream(dev), D.stream(dev)
    h2d_done = [D.event() for _ in range(2)]
    pass_done = [D.event() for _ in range(2)]

    def run_passes(srcs, j):
        o = 0
        for p in passes:
            eng.infer_batch(srcs[o:o + len(p)], det[j, o:o + len(p)], cnt[j, o:o + len(p)])
            o += len(p)
        det_h[j].copy_(det[j], non_blocking=True)
        cnt_h[j].copy_(cnt[j], non_blocking=True)
        if gatherer:
            # the one collective of the path, once per STEP and stream-ordered behind the step's kernels: this rank's NB
            # records, zero-padded to the busiest rank's count (a rank without frames still takes part).  Enqueue only -- the
            # buffers are preallocated and the per-rank trimming happens on the host after the timed region (no .item() here:
            # a host sync per step would serialise the CPU launch time of the next step behind this step's kernels)
            gathered[0] = gatherer.gather(det[j, :NB], cnt[j, :NB])

    gathered = [None]
    gatherer = shard.DetectionGatherer(rows, ncnt, NB_MAX, dev) if dist else None

    def step_resident(i, j):
        with D.use_stream(compute):
            run_passes(clouds, j)

    def step_host(i, j):
        # H2D of step i on the copy stream (waits until the pass that last read this staging set is done),
        # compute waits for it: the copy of step i+1 overlaps the kernels of step i
        s = i & 1
        with D.use_stream(copier):
            D.wait_event(copier, pass_done[s])
            for b in range(NB):
                stage_bufs[s][b].copy_(host[b], non_blocking=True)
            D.record(h2d_done[s], copier)
        with D.use_stream(compute):
            D.wait_event(compute, h2d_done[s])
            run_passes(stage_bufs[s], j)
            D.record(pass_done[s], compute)

    def timed_region(step, profile):
        for i in range(W):
            step(i, i % K)
        D.synchronize()
        if dist:
            dist.barrier()
        D.synchronize()
        if profile:
            eng.profile_begin()
        t0 = time.perf_counter()
        for i in range(K):
            step(i, i)
        D.synchronize()
        if dist:
            dist.barrier()
        D.synchronize()
        elapsed = time.perf_counter() - t0
        prof = eng.profile_end() if profile else None
        if dist:
            t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        return elapsed, prof

    for ev in pass_done:
        D.record(ev, compute)
    elapsed, prof = timed_region(step_resident, True)
    mean_det = float(cnt_h[:, :, 0].float().mean())
    elapsed_h, _ = timed_region(step_host, False)
    if gatherer:
        # after the timed regions: the last step's gathered records, trimmed per rank from host-known frame counts, must hold this
        # rank's own detections at its slot (the collective really moved them)
        n_all = args.global_batch if args.global_batch > 0 else world * NB
        counts = [len(shard.frames_for_rank(r, world, n_all)) for r in range(world)] if args.global_batch > 0 else [NB] * world
        g_det, g_cnt = gatherer.unpack(counts, gathered[0])
        assert sum(x.shape[0] for x in g_det) == n_all
        assert torch.equal(g_cnt[rank].cpu(), cnt[(K - 1) % K, :NB].cpu()) if NB else True

    if rank == 0:
        k_ms, k_n, k_flops = prof
        ratio = eng.executed_ratio()
        out = {
            "metric": f"point-cloud frames/sec end-to-end (voxelise→NMS), {args.config}, {world}x MI355X",  # BASELINE.json's metric (default config eight_20cm)
            "value": round(K * frames_per_step / elapsed, 3),
            "unit": "frames/s",
            "n_gpus": world,
            "steps": K,
            "warmup": W,
            "ms_per_step": round(elapsed / K * 1e3, 4),
            "higher_is_better": True,
            "scaling": scaling,
            "vs_baseline": None,
            "dtype": "f32" if args.precision == "fp32" else args.precision,
            "data": "synthetic",
            "value_host_start": round(K * frames_per_step / elapsed_h, 3),
            "ms_per_step_host_start": round(elapsed_h / K * 1e3, 4),
            "config": {"workload": f"configs/{args.config}.json, synthetic {CLOUD_DESC.get(args.config, 'LiDAR-shaped')} clouds "
                                   f"(mean {int(np.mean([c.shape[0] for c in clouds])) if clouds else 0} points, one distinct cloud per frame of a step), "
                                   "independent frames (batch=1 semantics), random-init weights (InstanceNorm backbone), AABB NMS; "
                                   "`value`: clouds resident in HBM; `value_host_start`: clouds in pinned host memory, H2D inside the timed region (SURVEY 8(d) timer)",
                       "frames_per_step": frames_per_step, "frames_per_pass_per_gpu": min(NB, MAXB), "engine_max_batch": ENG_B,
                       "parallelism": f"frame-sharded x{world}" + (", one all_gather of detection records per step" if world > 1 else ""),
                       "global_batch": args.global_batch or None, "cls_bias": args.cls_bias, "mean_detections": mean_det},
        }
        ach = (k_flops / (k_ms * 1e-3) / 1e12) if k_ms > 0 else 0.0
        traffic, traffic_src = hbm_traffic(eng.dominant_kernel(), min(NB, MAXB), eng.H, eng.W)
        kname = f"conv 3x3 s1 64->64 @ {eng.H}x{eng.W} x{min(NB, MAXB)} frames, tiling '{eng.dominant_kernel()}'"
        if args.precision == "fp32":
            out["roofline"] = {"bound": "mfma", "kernel": kname,
                               "achieved": round(ach * ratio, 3), "peak": F32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                               "frac": round(ach * ratio / F32_MFMA_PEAK_TFLOPS, 4), "traffic": traffic, "traffic_source": traffic_src,
                               "avg_launch_ms": round(k_ms, 5), "launches": k_n,
                               "algorithmic_flops_per_launch": k_flops, "executed_flops_per_launch": k_flops * ratio,
                               "algorithmic_tflops": round(ach, 3), "algorithmic_frac": round(ach / F32_MFMA_PEAK_TFLOPS, 4),
                               "note": "achieved / frac price the MFMA flops the kernel EXECUTES (hardware utilisation, <= 1): Winograd F(2x2,3x3) executes 4/9 of the "
                                       "direct-convolution count, F(4x4,3x3) 1/4 -- a FASTER kernel with fewer executed flops shows a LOWER frac; "
                                       "`algorithmic_tflops` / `algorithmic_frac` price SURVEY 8(d)'s direct-convolution flops against the same peak (> 1 for Winograd)" if ratio < 1.0 else "direct convolution: executed = algorithmic flops"}
        else:
            # 16-bit operands: the same layer is no longer bound by the matrix pipe.  Both floors are stated, the binding one
            # (the larger time) is the roofline: HBM with SURVEY 8(d)'s algorithmic bytes (input read once + output written
            # once: 2 * 4 * 64 * H * W per frame with fp32 activations, half that under fp16s), MFMA with the executed flops at the 16-bit dense peak.
            nfr = min(NB, MAXB)
            elem = 2 if args.precision == "fp16s" else 4  # fp16s stores its activation tensors in fp16
            alg_bytes = 2.0 * elem * 64 * eng.H * eng.W * nfr
            t_hbm = alg_bytes / (HBM_PEAK_GBS * 1e9) * 1e3
            t_mfma = k_flops * ratio / (LP_MFMA_PEAK_TFLOPS * 1e12) * 1e3
            hbm_bound = t_hbm >= t_mfma
            gbs = alg_bytes / (k_ms * 1e-3) / 1e9 if k_ms > 0 else 0.0
            out["roofline"] = {"bound": "hbm" if hbm_bound else "mfma", "kernel": kname,
                               "achieved": round(gbs, 1) if hbm_bound else round(ach * ratio, 2),
                               "peak": HBM_PEAK_GBS if hbm_bound else LP_MFMA_PEAK_TFLOPS, "unit": "GB/s" if hbm_bound else "TFLOP/s",
                               "frac": round((gbs / HBM_PEAK_GBS) if hbm_bound else (ach * ratio / LP_MFMA_PEAK_TFLOPS), 4),
                               "traffic": traffic, "traffic_source": traffic_src, "avg_launch_ms": round(k_ms, 5), "launches": k_n,
                               "algorithmic_bytes_per_launch": alg_bytes, "executed_flops_per_launch": k_flops * ratio,
                               "floors_ms": {"hbm_8TBps": round(t_hbm, 5), "mfma_16bit_2500TF": round(t_mfma, 5)},
                               "mfma_frac": round(ach * ratio / LP_MFMA_PEAK_TFLOPS, 4), "hbm_frac": round(gbs / HBM_PEAK_GBS, 4),
                               "algorithmic_tflops": round(ach, 2),  # useful (direct-convolution) flops: bf16x3 EXECUTES three MFMAs per product
                               "note": "16-bit operand mode: floors of this layer per launch in `floors_ms`; the larger one is the bound"}
        if args.precision != "fp32":
            out["tagged"] = (f"reduced-precision deploy mode '{args.precision}': convolutions, upsamplers and head on 16-bit MFMA operands, fp32 accumulation, "
                             + ("fp16 activation tensors in HBM" if args.precision == "fp16s" else "fp32 activations in HBM") +
                             " (SURVEY 8(f).4; the reference deploys TensorRT FP16 engines): NOT the headline -- "
                             "the headline is the fp32 line (dtype f32); tolerance table in DESIGN.md")
        if stub or (world > 1 and os.environ.get("PP_BENCH_BACKEND", "nccl") != "nccl"):
            out["rehearsal"] = "stub engine / ranks share devices, gather over gloo: not a measurement"
        if world == 1 and not args.no_extras and not stub:
            stages, whole, til = stage_rooflines(eng, clouds, min(NB, MAXB), cfg, mfma_peak=F32_MFMA_PEAK_TFLOPS if args.precision == "fp32" else LP_MFMA_PEAK_TFLOPS)
            out["roofline"]["stages"] = stages
            out["roofline"]["whole_frame"] = whole
            out["extras"] = extras(eng, eng_mod, synth, args, cfg, local, clouds, host, dev, D)
            out["extras"]["tilings"] = [L["tiling"] for L in til]
            out["roofline"]["batch1"] = out["extras"].pop("batch1_roofline")
        if world == 1 and not args.no_cpu_baseline and not stub:
            out["cpu_baseline"] = cpu_baseline(synth, args.config, args.cpu_frames)
        print(json.dumps(out), flush=True)
    if dist:
        dist.destroy_process_group()


def extras(eng, eng_mod, synth, args, cfg, local, clouds, host, dev, D):
    """Untimed side measurements (rank 0, N=1, after the timed regions)."""
    out = {}
    # batch=1 latency of this build (BASELINE config 2 is quoted at batch=1): one frame per pp_infer_frame call, synchronised
    # after every frame like train.py:236 -- resident cloud, and from pinned host memory
    # -- with an engine of its own (max_batch = 1: the tuner then measures the tilings at ONE frame per launch, where smaller
    # tiles win), as a batch-1 deployment would be built; `..._batch_engine` is the same through the 32-frame engine's tilings
    eng1 = eng_mod.Engine(dict(cfg), device_index=local, max_batch=1, precision=args.precision)
    eng1.load_state_dict(synth.seeded_state_dict(0, cls_bias=args.cls_bias, num_anchor_per_loc=eng1.num_anchor_per_loc))
    n = 30
    for tag, e in (("", eng1), ("_batch_engine", eng)):
        det1, cnt1 = e.infer_frame(clouds[0])
        torch.cuda.synchronize()
        for name, src in (("resident", clouds), ("host_start", host)):
            if tag and name == "host_start":
                continue
            t0 = time.perf_counter()
            for i in range(n):
                pts = src[i % len(src)]
                if name == "host_start":
                    pts = pts.to(dev, non_blocking=True)
                d, c = e.infer_frame(pts, det1, cnt1)
                c.cpu()
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            out[f"batch1_latency_ms_{name}{tag}"] = round(dt / n * 1e3, 4)
            out[f"batch1_frames_per_s_{name}{tag}"] = round(n / dt, 2)
    til1 = eng1.layer_tilings()
    out["batch1_tilings"] = [L["tiling"] for L in til1]
    # BASELINE config 2 (batch = 1) with its own roofline: executed MFMA flops of ONE frame over the whole synchronised call
    alg1, ex1 = frame_flops(eng1.H, eng1.W, til1)
    peak1 = F32_MFMA_PEAK_TFLOPS if args.precision == "fp32" else LP_MFMA_PEAK_TFLOPS
    ms1 = out["batch1_latency_ms_resident"]
    out["batch1_roofline"] = {"bound": "mfma", "ms_per_frame": ms1, "algorithmic_gflop": round(alg1 / 1e9, 2), "executed_gflop": round(ex1 / 1e9, 2),
                              "achieved": round(ex1 / (ms1 * 1e-3) / 1e12, 2), "peak": peak1, "unit": "TFLOP/s",
                              "frac": round(ex1 / (ms1 * 1e-3) / 1e12 / peak1, 4),
                              "note": "one pp_infer_frame per frame, host-synchronised after every frame (train.py:236); wall time incl. launch gaps"}
    del eng1
    # the same resident pass with a trained-like head bias (few candidates instead of ~900 detections per frame)
    if args.cls_bias is None:
        eng.load_state_dict(synth.seeded_state_dict(0, cls_bias=-4.6, num_anchor_per_loc=eng.num_anchor_per_loc))
        NB = min(len(clouds), eng.max_batch)
        for _ in range(2):
            d, c = eng.infer_batch(clouds[:NB])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            d, c = eng.infer_batch(clouds[:NB], d, c)
        torch.cuda.synchronize()
        out["value_trained_like_bias"] = round(10 * NB / (time.perf_counter() - t0), 3)
        out["mean_detections_trained_like_bias"] = float(c[:, 0].float().mean())
        eng.load_state_dict(synth.seeded_state_dict(0, num_anchor_per_loc=eng.num_anchor_per_loc))
    return out


if __name__ == "__main__":
    main()
