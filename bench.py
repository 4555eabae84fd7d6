#!/usr/bin/env python3
"""bench.py -- frames/s of the PointPillars hot path (voxelise -> NMS) on MI355X.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A step = one pass of `--batch` independent frames per GPU through pp_infer_batch (each frame keeps the
reference's batch=1 semantics, train.py:219-242: no statistic is shared between frames; the frames only share
kernel launches) plus the async D2H of their detection records.  Workload: configs/eight_20cm.json, synthetic KITTI-shape
20k-point clouds already resident in HBM, random-init weights of the reference architecture.
Frames are sharded by index across ranks (weak scaling: `--batch` frames per rank per step, no data-path
collective); RCCL only gathers the detection records once at the end of the timed region.
Rank 0 prints ONE JSON line with `roofline` (dominant conv kernel timed with HIP events on its launch
stream inside the timed region) and, at N=1, `cpu_baseline` (the CPU oracle on the host cores).
"""
import argparse
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


def hbm_traffic(tiling, frames):
    """HBM bytes per launch of the roofline kernel.  PMC counters cannot be read from inside this process, so the
    figure comes from the committed rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this same command
    (profiles/r01_hbm_traffic.json, made by tools/profile_round.sh + tools/hbm_traffic.py; FETCH_SIZE doubled per the
    gfx950 correction of MI355X_MICROARCH.md) -- and only if they were taken for the tiling and batch that ran."""
    path = os.path.join(ROOT, "profiles", "r01_hbm_traffic.json")
    try:
        with open(path) as f:
            t = json.load(f)
        e = t["tilings"].get(tiling)
        if e is None or t["frames_per_launch"] != frames:
            return None, None
        return int(e["hbm_bytes_per_launch"]), "profiles/r01_hbm_traffic.json (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, bytes/launch)"
    except (OSError, KeyError, ValueError):
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


def cpu_baseline(synth, n_frames=8):
    """The CPU oracle (oracle/: C for voxelise/mask/NMS, torch-CPU fp32 for PFN/backbone/head) on the
    host cores, bounded sample (1 warm + n_frames timed frames of the same workload)."""
    from oracle import c_oracle as C
    from oracle import pp_oracle as O
    cores = host_cores()
    torch.set_num_threads(cores)
    cfg = synth.load_config("eight_20cm")
    s = O.voxel_setup(cfg)
    a = O.make_anchors(s)
    sd = synth.seeded_state_dict(0)
    C.lib()
    clouds = [synth.lidar_cloud("eight_20cm", seed=1000 + i) for i in range(n_frames + 1)]

    def frame(pts):
        v, c, n = C.points_to_voxels(pts, s["voxel_size"], s["offset"], s["grid_size"], cfg["max_voxels"], cfg["max_num_points"])
        mask = C.create_mask(c, s["grid_size"], a["anchors_coors"])
        rpn = O.backbone(O.scatter(O.pfn(v, n, c, sd, s), c, s["grid_size"]), sd)
        cls, box, dr = O.head(rpn, sd)
        return O.postprocess(cls, box, dr, mask, a["anchors"], a["class_masks"], cfg["center_limit"])

    frame(clouds[0])
    t0 = time.time()
    for p in clouds[1:]:
        frame(p)
    dt = time.time() - t0
    return {"value": round(n_frames / dt, 4), "unit": "frames/s", "cores": cores, "kind": "port",
            "sample": f"{n_frames} frames of eight_20cm (20k-pt clouds, batch=1) after 1 warm-up, oracle/ CPU path, {dt:.1f} s"}


def extras(eng, clouds, dev, NB, rows):
    """Untimed side measurements for the report (rank 0, N=1, after the timed region):
    * the PCIe-inclusive rate: the same pass with the clouds in pinned HOST memory, their H2D copies enqueued ahead
      of pp_infer_batch on the same stream (SURVEY 8(d)'s timer; never `value`);
    * single-frame latency of the stand-alone stage entry points in the reference's buckets (train.py:224-236):
      pre = voxelise + anchor mask, net = PFN + scatter + backbone + head, post = post-processing."""
    out = {}
    host = [c.cpu().pin_memory() for c in clouds]
    stage = [[torch.empty_like(c) for c in clouds[:1] * NB] for _ in range(2)]
    for b in range(NB):
        for k in range(2):
            stage[k][b] = torch.empty_like(clouds[b % len(clouds)])
    det = torch.zeros((NB, rows, 9), dtype=torch.float32, device=dev)
    cnt = torch.zeros((NB, 1 + 8), dtype=torch.int32, device=dev)
    det_h = torch.zeros((NB, rows, 9), dtype=torch.float32).pin_memory()
    cnt_h = torch.zeros((NB, 1 + 8), dtype=torch.int32).pin_memory()

    def hstep(i):
        bufs = stage[i & 1]
        for b in range(NB):
            bufs[b].copy_(host[b % len(host)], non_blocking=True)
        eng.infer_batch(bufs, det, cnt)
        det_h.copy_(det, non_blocking=True)
        cnt_h.copy_(cnt, non_blocking=True)

    for i in range(2):
        hstep(i)
    torch.cuda.synchronize()
    n = 10
    t0 = time.perf_counter()
    for i in range(n):
        hstep(i)
    torch.cuda.synchronize()
    out["value_with_h2d"] = round(n * NB / (time.perf_counter() - t0), 3)

    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    acc = [0.0, 0.0, 0.0]
    reps = 5
    for r in range(reps + 1):
        pts = clouds[r % len(clouds)]
        ev[0].record()
        vox, coors, npts, num = eng.voxelize(pts)
        mask = eng.anchor_mask(coors, num)
        ev[1].record()
        cls, box, dr = eng.head(eng.backbone(eng.scatter(eng.pfn(vox, coors, npts, num), coors, num)))
        ev[2].record()
        eng.postprocess(cls, box, dr, mask)
        ev[3].record()
        torch.cuda.synchronize()
        if r:  # first repetition warms up
            for k in range(3):
                acc[k] += ev[k].elapsed_time(ev[k + 1])
    out["stage_ms_single_frame"] = {"pre": round(acc[0] / reps, 4), "net": round(acc[1] / reps, 4), "post": round(acc[2] / reps, 4),
                                    "note": "stand-alone stage entry points, one frame, dense canvas (the fused batched path shares launches across frames)"}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--config", default="eight_20cm")
    ap.add_argument("--cls-bias", type=float, default=None, help="'trained-like' head bias (e.g. -4.6); default random init")
    ap.add_argument("--streams", type=int, default=1,
                    help="frames in flight per GPU: independent frames run on separate HIP streams (one pp_ctx each) so one "
                         "frame's kernel tails / small kernels overlap another frame's MFMA work")
    ap.add_argument("--batch", type=int, default=32,
                    help="independent frames per pass on one stream (pp_infer_batch: frame = grid.z of the conv launches)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-frames", type=int, default=20)
    ap.add_argument("--no-extras", action="store_true", help="skip the untimed side measurements (PCIe-inclusive rate, stage latencies)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # PP_BENCH_BACKEND=gloo: rehearsal of the N-rank flow on a box with fewer GPUs than ranks (ranks share devices,
        # the gather goes through the host) -- RCCL refuses two ranks on one device.  Never a valid measurement.
        backend = os.environ.get("PP_BENCH_BACKEND", "nccl")
        if backend != "nccl":
            local = local % torch.cuda.device_count()
        torch.cuda.set_device(local)
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)

    synth = importlib.import_module("3d_object_detection_amd.synth")
    eng_mod = importlib.import_module("3d_object_detection_amd.engine")
    shard = importlib.import_module("3d_object_detection_amd.shard")
    cfg = synth.load_config(args.config)
    cfg["device"] = dev
    S = max(1, args.streams)
    engines, streams = [], []
    for _ in range(S):
        e = eng_mod.Engine(dict(cfg), device_index=local, max_batch=max(1, args.batch))
        e.load_state_dict(synth.seeded_state_dict(0, cls_bias=args.cls_bias))
        engines.append(e)
        streams.append(torch.cuda.Stream(device=dev))
    eng = engines[0]

    pool = 8
    clouds = [torch.from_numpy(synth.lidar_cloud(args.config, seed=1000 + rank * pool + i)).to(dev) for i in range(pool)]
    K, W = args.steps, args.warmup
    NB = max(1, args.batch)
    rows = eng.cfg.num_classes * eng.cfg.nms_post_max
    det = torch.zeros((K, NB, rows, 9), dtype=torch.float32, device=dev)
    cnt = torch.zeros((K, NB, 1 + 8), dtype=torch.int32, device=dev)
    det_h = torch.zeros((K, NB, rows, 9), dtype=torch.float32).pin_memory()
    cnt_h = torch.zeros((K, NB, 1 + 8), dtype=torch.int32).pin_memory()

    def step(i, j):
        # one step = NB independent frames (batch=1 semantics per frame: no cross-frame statistics)
        with torch.cuda.stream(streams[i % S]):
            engines[i % S].infer_batch([clouds[(i * NB + b) % pool] for b in range(NB)], det[j], cnt[j])
            det_h[j].copy_(det[j], non_blocking=True)
            cnt_h[j].copy_(cnt[j], non_blocking=True)

    for i in range(W):
        step(i, i % K)
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    for e in engines:
        e.profile_begin()
    t0 = time.perf_counter()
    for i in range(K):
        step(i, i)
    if dist:
        for s_ in streams:  # the one collective of the path follows the compute it gathers (the steps ran on side streams)
            torch.cuda.current_stream(dev).wait_stream(s_)
        shard.gather_detections(det.view(K * NB, rows, 9), cnt.view(K * NB, 1 + 8))
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    prof = [e.profile_end() for e in engines]
    k_n = sum(q[1] for q in prof)
    k_ms = sum(q[0] * q[1] for q in prof) / max(k_n, 1)
    k_flops = prof[0][2]
    if dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        out = {
            "metric": f"point-cloud frames/sec end-to-end (voxelise→NMS), {args.config}, 1/2/4/8 MI355X",  # BASELINE.json's metric (default config eight_20cm)
            "value": round(world * K * NB / elapsed, 3),
            "unit": "frames/s",
            "n_gpus": world,
            "steps": K,
            "warmup": W,
            "ms_per_step": round(elapsed / K * 1e3, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"configs/{args.config}.json, synthetic KITTI-shape 20k-point clouds resident in HBM, "
                                   "independent frames (batch=1 semantics, NB per pass), random-init weights (InstanceNorm backbone), AABB NMS",
                       "frames_per_step": world * NB, "frames_per_pass_per_gpu": NB, "parallelism": f"frame-sharded x{world}", "streams_per_gpu": S,
                       "cls_bias": args.cls_bias, "mean_detections": float(cnt_h[:, :, 0].float().mean())},
        }
        ach = (k_flops / (k_ms * 1e-3) / 1e12) if k_ms > 0 else 0.0
        traffic, traffic_src = hbm_traffic(eng.dominant_kernel(), NB)
        out["roofline"] = {"bound": "mfma", "kernel": f"conv 3x3 s1 64->64 @ {eng.H}x{eng.W} x{NB} frames, tiling '{eng.dominant_kernel()}'",
                           "achieved": round(ach, 3), "peak": F32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                           "frac": round(ach / F32_MFMA_PEAK_TFLOPS, 4), "traffic": traffic, "traffic_source": traffic_src,
                           "avg_launch_ms": round(k_ms, 5), "launches": k_n, "flops_per_launch": k_flops}
        if eng.dominant_kernel().startswith("wino"):
            # Winograd F(2x2,3x3) issues 16 MFMA multiplications where the direct form needs 36: `achieved` prices the
            # ALGORITHMIC (direct-conv) flops and can pass the MFMA peak; `executed` is what the matrix cores really ran
            out["roofline"]["algorithm"] = "Winograd F(2x2,3x3): executed MFMA flops = algorithmic x 4/9"
            out["roofline"]["executed"] = round(ach * 4.0 / 9.0, 3)
            out["roofline"]["executed_frac"] = round(ach * 4.0 / 9.0 / F32_MFMA_PEAK_TFLOPS, 4)
        if world > 1 and os.environ.get("PP_BENCH_BACKEND", "nccl") != "nccl":
            out["rehearsal"] = "ranks share devices, gather over " + os.environ["PP_BENCH_BACKEND"] + ": not a measurement"
        if world == 1 and not args.no_extras:
            out["extras"] = extras(eng, clouds, dev, NB, rows)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(synth, args.cpu_frames)
        print(json.dumps(out), flush=True)
    if dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
