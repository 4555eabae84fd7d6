"""Frame sharding across the GPUs of one node.

Frames are independent (no cross-frame statistics), so a batch is sharded by frame index with no
data-path collective; the only exchange is ONE all_gather of fixed-size detection records -- per frame
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
    return torch.cat([det.reshape(f, -1), cnt.contiguous().view(torch.float32).reshape(f, -1)], dim=1).contiguous()


def unpack_records(rec, rows, ncnt):
    f = rec.shape[0]
    det = rec[:, :rows * 9].reshape(f, rows, 9)
    cnt = rec[:, rows * 9:rows * 9 + ncnt].contiguous().view(torch.int32)
    return det, cnt


def gather_detections(det, cnt, group=None):
    """det f32[F, rows, 9], cnt i32[F, 1+C] for this rank's F frames -> lists over ranks (same shapes).
    One collective of fixed-size padded records (latency-bound: ~32 KB per frame)."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        return [det], [cnt]
    rec = pack_records(det, cnt)
    out = torch.empty((world,) + tuple(rec.shape), dtype=rec.dtype, device=rec.device)
    dist.all_gather(list(out.unbind(0)), rec, group=group)  # one collective; views of one buffer (gloo rejects the _into_tensor form's shape)
    pairs = [unpack_records(out[r], det.shape[1], cnt.shape[1]) for r in range(world)]
    return [p[0] for p in pairs], [p[1] for p in pairs]


def share_tuning(lib, group=None, src=0):
    """Rank `src` has created its engine (the tuner ran there); every other rank imports its table BEFORE creating
    its own, so all ranks run identical kernels (same speed, bit-identical detections to the 1-GPU run)."""
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
