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


class DetectionGatherer:
    """The per-step exchange of a timed loop: buffers allocated ONCE, `gather()` only enqueues (pack into the preallocated
    block, one all_gather) -- no `.item()`, no host-side trimming, nothing that would drain the stream before the next step's
    launches are enqueued.  Every rank pads its records with zero frames to `pad_to` (the busiest rank's count); how many
    frames each rank really holds is host knowledge (`frames_for_rank`), so `unpack()` trims AFTER the timed region from
    `counts`.  `force_collective` runs the collective even for a single rank (the RCCL smoke test on a one-GPU box)."""

    def __init__(self, rows, ncnt, pad_to, device, group=None, force_collective=False):
        self.group, self.rows, self.ncnt, self.pad_to = group, rows, ncnt, pad_to
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.collective = dist.is_initialized() and (self.world > 1 or force_collective)
        width = rows * 9 + ncnt
        self.block = torch.zeros((max(pad_to, 1), width), dtype=torch.float32, device=device)
        self.out = torch.zeros((self.world, max(pad_to, 1), width), dtype=torch.float32, device=device)
        self._views = list(self.out.unbind(0))  # views of one buffer (gloo rejects the _into_tensor form's shape)

    def gather(self, det, cnt):
        """det f32[F, rows, 9], cnt i32[F, ncnt] (F <= pad_to) -> f32[world, pad_to, rows*9 + ncnt], stream-ordered."""
        f = det.shape[0]
        if f > self.pad_to:
            raise ValueError(f"DetectionGatherer: {f} frames on this rank exceed pad_to = {self.pad_to}")
        if f:
            self.block[:f, :self.rows * 9].copy_(det.reshape(f, self.rows * 9))
            self.block[:f, self.rows * 9:].copy_(cnt.contiguous().view(torch.float32).reshape(f, self.ncnt))
        if self.collective:
            dist.all_gather(self._views, self.block, group=self.group)
        else:
            self.out[0].copy_(self.block)
        return self.out

    def unpack(self, counts, out=None):
        """Host side, after the loop: per-rank (det[F_r, rows, 9], cnt[F_r, ncnt]) lists trimmed to counts[r] frames."""
        out = self.out if out is None else out
        dets, cnts = [], []
        for r in range(self.world):
            d, c = unpack_records(out[r, :counts[r]], self.rows, self.ncnt)
            dets.append(d)
            cnts.append(c)
        return dets, cnts


def gather_detections(det, cnt, group=None, pad_to=None, counts=None, force_collective=False):
    """One-shot convenience form (NOT for a timed loop: it allocates, and without `counts` it synchronises the host).
    det f32[F, rows, 9], cnt i32[F, 1+C] for this rank's F frames -> lists over ranks, each trimmed to that rank's own F.
    Ranks may hold DIFFERENT frame counts (a global batch not divisible by the world size, or a rank with no frame at all): every
    rank pads its records with zero frames to `pad_to` (default: the maximum over ranks).  `counts` = the frames of every rank when
    the caller knows them (frames_for_rank); otherwise they are exchanged with one extra 8-byte all_gather and read on the host."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1 and not (force_collective and dist.is_initialized()):
        return [det], [cnt]
    f = det.shape[0]
    if counts is None:
        fs = [torch.zeros(1, dtype=torch.int64, device=det.device) for _ in range(world)]
        dist.all_gather(fs, torch.tensor([f], dtype=torch.int64, device=det.device), group=group)
        counts = [int(x.item()) for x in fs]
    if pad_to is None:
        pad_to = max(counts)
    g = DetectionGatherer(det.shape[1], cnt.shape[1], pad_to, det.device, group=group, force_collective=force_collective)
    return g.unpack(counts, g.gather(det, cnt))


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
