"""Thin device layer bench.py drives (streams, events, pinned memory on the HIP device).  The CPU rehearsal of the
N-rank flow (tests/bench_stub.py) offers the same functions over no-ops, so the rank logic of bench.py runs unchanged."""
import contextlib

import torch


def pin(t):
    return t.pin_memory()


def stream(dev):
    return torch.cuda.Stream(device=dev)


def current_stream(dev):
    return torch.cuda.current_stream(dev)


def event():
    return torch.cuda.Event()


def record(ev, st):
    ev.record(st)


def wait_event(st, ev):
    st.wait_event(ev)


def wait_stream(st, other):
    st.wait_stream(other)


def use_stream(st):
    return torch.cuda.stream(st)


def synchronize():
    torch.cuda.synchronize()
