"""Frame sharding across the GPUs of one node.

Frames are independent (no cross-frame statistics), so a batch is sharded by frame index with no
data-path collective; the only exchange is one all_gather per step of fixed-size detection records -- per frame
`rows * 9` floats of boxes followed by the `1 + C` int32 counts bit-cast into the same float record --
over RCCL/xGMI on GPUs ("nccl" backend), gloo in the CPU tests."""
import torch
import torch.distributed as dist


def frames_for_rank(rank, world_size, n_frames):
    """Round-robin: rank g owns frames {i : i mod world_size == g}."""
    return list(range(rank, n_frames, world_size))


def pack_records(det, cnt):
    """det f32[F, rows, 9], cnt i32[F, 1+C] -> one f32[F, rows*9 + 1+C] record per frame (counts bit-cast)."""
    f = det.shape[0]
    return torch.cat([det.reshape(f, det.shape[1] * 9), cnt.contiguous().view(torch.float32).reshape(f, cnt.shape[1])], dim=1).contiguous()


def unpack_records(rec, rows, ncnt):
    f = rec.shape[0]
    det = rec[:, :rows * 9].reshape(f, rows, 9)
    cnt = rec[:, rows * 9:rows * 9 + ncnt].contiguous().view(torch.int32)
    return det, cnt


def frames_per_rank_max(world_size, n_frames):
    """Frames of the busiest rank under frames_for_rank: every rank sizes its engine (max_batch) and pads its records to this."""
    return (n_frames + world_size - 1) // world_size


def gather_detections(det, cnt, group=None, pad_to=None):
    """det f32[F, rows, 9], cnt i32[F, 1+C] for this rank's F frames -> lists over ranks, each trimmed to that rank's own F.
    One collective of fixed-size records (latency-bound: ~32 KB per frame).  Ranks may hold DIFFERENT frame counts (a global
    batch not divisible by the world size, or a rank with no frame at all): every rank pads its records with zero frames to
    `pad_to` (default: the maximum over ranks, found with one extra 8-byte all_gather) and appends its true F to the record
    block, so the receiver can trim."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        return [det], [cnt]
    f, rows, ncnt = det.shape[0], det.shape[1], cnt.shape[1]
    if pad_to is None:
        fs = [torch.zeros(1, dtype=torch.int64, device=det.device) for _ in range(world)]
        dist.all_gather(fs, torch.tensor([f], dtype=torch.int64, device=det.device), group=group)
        pad_to = max(int(x.item()) for x in fs)
    if f > pad_to:
        raise ValueError(f"gather_detections: {f} frames on this rank exceed pad_to = {pad_to}")
    width = rows * 9 + ncnt
    block = torch.zeros((pad_to + 1, width), dtype=torch.float32, device=det.device)
    if f:
        block[:f] = pack_records(det, cnt)
    block[pad_to, 0] = float(f)  # exact for any frame count that fits a pass
    out = torch.empty((world,) + tuple(block.shape), dtype=block.dtype, device=block.device)
    dist.all_gather(list(out.unbind(0)), block, group=group)  # one collective; views of one buffer (gloo rejects the _into_tensor form's shape)
    dets, cnts = [], []
    for r in range(world):
        fr = int(out[r, pad_to, 0].item())
        d, c = unpack_records(out[r, :fr], rows, ncnt)
        dets.append(d)
        cnts.append(c)
    return dets, cnts


def share_tuning(lib, group=None, src=0):
    """Rank `src` has created its engine (the tuner ran there); every other rank imports its table BEFORE creating
    its own, so all ranks run identical kernels when their engines are built for the same max_batch (the tuner's key and the
    Winograd strip decision carry it): same speed, and a frame's detections are then the ones a 1-GPU engine of that
    max_batch produces, whatever rank and position in the pass it lands on."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return 0
    import ctypes
    if dist.get_rank(group) == src:
        n = lib.pp_tune_export(None, 0)
        buf = ctypes.create_string_buffer(n + 1)
        lib.pp_tune_export(buf, n + 1)
        obj = [buf.value.decode()]
    else:
        obj = [None]
    dist.broadcast_object_list(obj, src=src, group=group)
    if dist.get_rank(group) != src:
        return lib.pp_tune_import(obj[0].encode())
    return 0


def merge_in_frame_order(dets, cnts, n_frames):
    """Undo the round-robin: returns per-frame (det rows, counts) in global frame order."""
    world = len(dets)
    out = [None] * n_frames
    for r in range(world):
        for j, f in enumerate(range(r, n_frames, world)):
            k = int(cnts[r][j, 0])
            out[f] = (dets[r][j, :k], cnts[r][j])
    return out
