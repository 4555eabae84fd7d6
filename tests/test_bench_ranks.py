"""CPU: bench.py's own N-rank flow (rank environment, rank 0 tunes and broadcasts its table, sharding, the two timed
regions, the single all_gather, MAX over ranks, rank 0's one JSON line) under torch.distributed.run with gloo and a
stubbed engine (tests/bench_stub.py).  The GPU run executes the same file over RCCL."""
import json
import os
import socket
import subprocess
import sys

import pytest

from conftest import ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def run_bench(nproc, extra):
    env = dict(os.environ)
    env["PP_BENCH_ENGINE"] = "bench_stub"
    env["PYTHONPATH"] = os.path.join(ROOT, "tests") + os.pathsep + env.get("PYTHONPATH", "")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", str(nproc), "--steps", "3", "--warmup", "1"] + extra
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout  # rank 0 prints ONE JSON line, the other ranks none
    return json.loads(lines[0])


def test_weak_scaling_rank_flow():
    out = run_bench(2, ["--batch", "4"])
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and out["steps"] == 3 and out["warmup"] == 1
    assert out["config"]["frames_per_step"] == 8 and out["config"]["frames_per_pass_per_gpu"] == 4
    assert out["value"] > 0 and out["value_host_start"] > 0 and out["unit"] == "frames/s"
    assert abs(out["value"] - 3 * 8 / (out["ms_per_step"] * 3e-3)) / out["value"] < 1e-3
    assert out["roofline"]["frac"] <= 1.0 and "rehearsal" in out
    # rank 0 tuned, and printed the line: its own table, not an imported one
    assert "tuned on this rank" in out["roofline"]["kernel"]


def test_global_batch_strong_scaling_flow():
    out = run_bench(2, ["--global-batch", "6"])
    assert out["scaling"] == "strong" and out["config"]["frames_per_step"] == 6 and out["config"]["global_batch"] == 6
    assert out["config"]["frames_per_pass_per_gpu"] == 3


def test_global_batch_not_divisible_and_idle_rank():
    """a global batch the world size does not divide (5 = 3 + 2) and one that leaves a rank without frames (1):
    every rank builds its engine for the busiest rank's frame count and pads its records, the per-step gather completes."""
    out = run_bench(2, ["--global-batch", "5"])
    assert out["config"]["frames_per_step"] == 5 and out["config"]["engine_max_batch"] == 3 and out["value"] > 0
    out = run_bench(2, ["--global-batch", "1"])
    assert out["config"]["frames_per_step"] == 1 and out["config"]["engine_max_batch"] == 1 and out["value"] > 0


def test_tuning_is_shared_not_repeated():
    """share_tuning: a rank that is not the source imports the source's table before it builds its engine."""
    import importlib
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    stub = importlib.import_module("bench_stub")
    stub._TUNE.clear()
    lib = stub.tuning_lib()
    e0 = stub.Engine({})
    e0.load_state_dict({})
    n = lib.pp_tune_export(None, 0)
    import ctypes
    buf = ctypes.create_string_buffer(n + 1)
    lib.pp_tune_export(buf, n + 1)
    stub._TUNE.clear()
    assert lib.pp_tune_import(buf.value) == 1
    e1 = stub.Engine({})
    e1.load_state_dict({})
    assert e0.tuned_here and not e1.tuned_here and "imported" in e1.dominant_kernel()


def test_pmc_traffic_only_for_the_profiled_build_and_shape():
    """roofline.traffic comes from a committed rocprofv3 --pmc pass: bench.py may quote it only for the layer shape, tiling,
    batch and HIP source tree it was taken on (sha256 of csrc/*.hip + *.h), otherwise null."""
    import importlib.util, json
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    files = sorted(f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_hbm_traffic.json"))
    assert files
    t = json.load(open(os.path.join(ROOT, "profiles", files[-1])))
    tiling = next(iter(t["tilings"]))
    h, w = (int(v) for v in t["layer"].split("@")[1].split("x"))
    got, src = bench.hbm_traffic(tiling, t["frames_per_launch"], h, w)
    if t.get("source_hash") == bench.source_hash():   # the committed figure belongs to this very tree
        assert got == int(t["tilings"][tiling]["hbm_bytes_per_launch"]) and files[-1] in src
    else:
        assert got is None
    assert bench.hbm_traffic(tiling, t["frames_per_launch"], h // 2, w)[0] is None          # another layer shape
    assert bench.hbm_traffic(tiling, t["frames_per_launch"] + 1, h, w)[0] is None           # another batch
    assert bench.hbm_traffic("no such tiling", t["frames_per_launch"], h, w)[0] is None
