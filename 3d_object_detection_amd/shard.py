"""Frame sharding across the GPUs of one node.

Frames are independent (no cross-frame statistics), so a batch is sharded by frame index with no
data-path collective; the only exchange is one all_gather of the fixed-size detection records
(det f32[rows,9] + counts) -- over RCCL/xGMI on GPUs ("nccl" backend), gloo in the CPU tests."""
import torch
import torch.distributed as dist


def frames_for_rank(rank, world_size, n_frames):
    """Round-robin: rank g owns frames {i : i mod world_size == g}."""
    return list(range(rank, n_frames, world_size))


def gather_detections(det, cnt, group=None):
    """det f32[F, rows, 9], cnt i32[F, 1+C] for this rank's F frames -> lists over ranks (same shapes).
    Single collective, fixed-size padded records (latency-bound: ~32 KB per frame)."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        return [det], [cnt]
    dets = [torch.empty_like(det) for _ in range(world)]
    cnts = [torch.empty_like(cnt) for _ in range(world)]
    dist.all_gather(dets, det, group=group)
    dist.all_gather(cnts, cnt, group=group)
    return dets, cnts


def merge_in_frame_order(dets, cnts, n_frames):
    """Undo the round-robin: returns per-frame (det rows, counts) in global frame order."""
    world = len(dets)
    out = [None] * n_frames
    for r in range(world):
        for j, f in enumerate(range(r, n_frames, world)):
            k = int(cnts[r][j, 0])
            out[f] = (dets[r][j, :k], cnts[r][j])
    return out
