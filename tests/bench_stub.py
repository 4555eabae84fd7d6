"""Stub engine + device layer for the CPU rehearsal of bench.py's rank flow (tests/test_bench_ranks.py,
PP_BENCH_ENGINE=bench_stub): same surface as 3d_object_detection_amd.engine.Engine / .devlayer, CPU tensors,
detections that encode (frame size, rank) so the gather can be checked.  TEST INFRASTRUCTURE ONLY."""
import contextlib
import ctypes
import types

import torch

_TUNE = {}


def device():
    return torch.device("cpu")


def pin(t):
    return t


def stream(dev):
    return object()


def current_stream(dev):
    return object()


def event():
    return object()


def record(ev, st):
    pass


def wait_event(st, ev):
    pass


def wait_stream(st, other):
    pass


def use_stream(st):
    return contextlib.nullcontext()


def synchronize():
    pass


class _TuneLib:
    """pp_tune_export / pp_tune_import over a dict: rank 0 'tunes' (records a table), the others must import it."""

    def pp_tune_export(self, buf, cap):
        t = "".join(f"{k}\t{v}\n" for k, v in sorted(_TUNE.items())).encode()
        if buf is not None and cap > 0:
            ctypes.memmove(buf, t + b"\0", min(cap, len(t) + 1))
        return len(t)

    def pp_tune_import(self, text):
        n = 0
        for line in text.decode().splitlines():
            k, v = line.split("\t")
            _TUNE[k] = v
            n += 1
        return n


def tuning_lib():
    return _TuneLib()


class Engine:
    def __init__(self, config, device_index=0, max_batch=1, **kw):
        self.cfg = types.SimpleNamespace(num_classes=3, nms_post_max=4)
        self.cnt_stride = 9
        self.max_batch = max_batch
        self.H = self.W = 8
        self.tuned_here = False
        self.imported = dict(_TUNE)

    def load_state_dict(self, sd):
        if not _TUNE:  # "autotune": only a rank that received no table measures its own
            _TUNE["k0 s1 c64"] = "wino stub"
            self.tuned_here = True

    def infer_batch(self, clouds, det, cnt, nms_mode=0):
        for b, c in enumerate(clouds):
            det[b].fill_(float(c.shape[0]))
            cnt[b].zero_()
            cnt[b, 0] = c.shape[0] % 7
        return det, cnt

    def profile_begin(self):
        pass

    def profile_end(self):
        return 1.0, 3, 1e9

    def executed_ratio(self):
        return 4.0 / 9.0

    def dominant_kernel(self):
        return _TUNE.get("k0 s1 c64", "untuned") + (" (tuned on this rank)" if self.tuned_here else " (imported)")
