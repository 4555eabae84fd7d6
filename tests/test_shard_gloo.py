"""CPU: the N>1 frame-sharding path with world_size=2 over gloo (the GPU run uses the same code over RCCL)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT, load_pkg


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_frames, out):
    import importlib
    import sys
    sys.path.insert(0, ROOT)
    shard = importlib.import_module("3d_object_detection_amd.shard")
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mine = shard.frames_for_rank(rank, world, n_frames)
    rows = 6
    det = torch.zeros((len(mine), rows, 9))
    cnt = torch.zeros((len(mine), 4), dtype=torch.int32)
    for j, f in enumerate(mine):  # fake per-frame detections: frame f has (f % rows) rows filled with f
        k = f % rows
        det[j, :k] = float(f)
        cnt[j, 0] = k
    t0 = torch.tensor([0.1 * (rank + 1)], dtype=torch.float64)
    dist.all_reduce(t0, op=dist.ReduceOp.MAX)  # the max-over-ranks timing of bench.py
    # the documented pad (every rank can compute it) and the discovered one must agree
    dets, cnts = shard.gather_detections(det, cnt, pad_to=shard.frames_per_rank_max(world, n_frames))
    dets2, cnts2 = shard.gather_detections(det, cnt)
    same = all(torch.equal(a, b) for a, b in zip(dets, dets2)) and all(torch.equal(a, b) for a, b in zip(cnts, cnts2))
    same &= [d.shape[0] for d in dets] == [len(shard.frames_for_rank(r, world, n_frames)) for r in range(world)]
    # the timed loop's form: buffers allocated once, gather() only enqueues, trimming from host-known counts afterwards;
    # two steps through the same buffers (the second with other values must replace the first's)
    counts = [len(shard.frames_for_rank(r, world, n_frames)) for r in range(world)]
    g = shard.DetectionGatherer(rows, 4, shard.frames_per_rank_max(world, n_frames), det.device)
    g.gather(det * 0.5, cnt)
    d3, c3 = g.unpack(counts, g.gather(det, cnt))
    same &= all(torch.equal(a, b) for a, b in zip(dets, d3)) and all(torch.equal(a, b) for a, b in zip(cnts, c3))
    merged = shard.merge_in_frame_order(dets, cnts, n_frames)
    ok = abs(float(t0) - 0.1 * world) < 1e-12 and same
    for f, (d, c) in enumerate(merged):
        ok &= d.shape[0] == f % rows and bool((d == float(f)).all()) and int(c[0]) == f % rows
    out[rank] = bool(ok)
    dist.destroy_process_group()


def test_frame_sharding_world2():
    shard = load_pkg("shard")
    assert shard.frames_for_rank(0, 2, 5) == [0, 2, 4] and shard.frames_for_rank(1, 2, 5) == [1, 3]
    assert sorted(sum((shard.frames_for_rank(r, 8, 64) for r in range(8)), [])) == list(range(64))
    mgr = mp.Manager()
    out = mgr.dict()
    port = _free_port()
    mp.spawn(_worker, args=(2, port, 8, out), nprocs=2, join=True)
    assert out[0] and out[1]


def test_unequal_and_empty_shards_world2():
    """A global batch that the world size does not divide (5 frames: 3 + 2) and one smaller than the world (1 frame: rank 1
    holds nothing): the padded gather must deliver every frame and trim every rank to its own count."""
    for n_frames in (5, 1):
        mgr = mp.Manager()
        out = mgr.dict()
        mp.spawn(_worker, args=(2, _free_port(), n_frames, out), nprocs=2, join=True)
        assert out[0] and out[1], n_frames


def test_single_process_gather_is_identity():
    shard = load_pkg("shard")
    det, cnt = torch.ones((2, 3, 9)), torch.tensor([[3, 1, 1, 1], [2, 1, 1, 0]], dtype=torch.int32)
    d, c = shard.gather_detections(det, cnt)
    assert len(d) == 1 and d[0] is det and c[0] is cnt


def test_force_collective_single_rank_gloo():
    """world_size 1 with force_collective: the pack -> all_gather -> unpack code really runs (the -m gpu twin of this test does
    the same over RCCL on device tensors, tests/test_gpu_parity.py::test_rccl_single_rank_gather)."""
    shard = load_pkg("shard")
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(_free_port())
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        det = torch.arange(2 * 3 * 9, dtype=torch.float32).reshape(2, 3, 9)
        cnt = torch.tensor([[3, 1, 1, 1], [2, 1, 1, 0]], dtype=torch.int32)
        d, c = shard.gather_detections(det, cnt, pad_to=4, counts=[2], force_collective=True)
        assert len(d) == 1 and d[0] is not det and torch.equal(d[0], det) and torch.equal(c[0], cnt)
    finally:
        dist.destroy_process_group()
