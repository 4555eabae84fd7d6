"""Static check for wino6_mfma (csrc/wino6.hip): no COPY of a register an asm request may still be writing.
The kernel issues its vector-memory requests from asm and waits with its own s_waitcnt counts, so hipcc believes a destination VGPR holds its
value from the request on.  If the register allocator then moves that value (v_mov_b32 / v_mov_b64 / v_accvgpr_write / a scratch spill) --
typically to reconcile two register assignments where control flow merges -- the copy takes stale data when the request has not landed yet.
This script compiles wino6.hip to ISA with the Makefile's flags and lists, per kernel, every such instruction whose SOURCE is a VGPR that some
buffer_load_dword* of the kernel writes.  Expected output: none (round 4 found two v_mov_b64 pairs at a loop exit this way).
  python tools/inflight_check.py [path/to/wino6.s]"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1:
    asm = open(sys.argv[1]).read()
else:
    out = os.path.join(tempfile.mkdtemp(), "wino6.s")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=off", "-fno-slp-vectorize",
                           "-munsafe-fp-atomics", "-S", "--cuda-device-only", os.path.join(ROOT, "3d_object_detection_amd", "csrc", "wino6.hip"), "-o", out],
                          stderr=subprocess.DEVNULL)
    asm = open(out).read()
lines = asm.split("\n")
def regs(tok):
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m: return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r"v(\d+)", tok)
    return {int(m.group(1))} if m else set()
kernels, cur = [], None
for i, l in enumerate(lines):
    m = re.match(r"^(_Z\w*wino6_mfma\w*):", l)
    if m: cur = [m.group(1), i, None]; kernels.append(cur)
    if cur and cur[2] is None and "s_endpgm" in l: cur[2] = i
bad_total = 0
for name, lo, hi in kernels:
    # in text order (an approximation of control flow): a request makes its destination registers "pending"; the first instruction that
    # READS a pending register other than a copy consumes it (the kernel waits in front of every use); a copy / spill of a pending
    # register is reported
    pending, dests, bad = set(), set(), []
    for i in range(lo, hi):
        txt = lines[i].split(";")[0].strip()
        t = txt.replace(",", " ").split()
        if not t or t[0].endswith(":") or t[0].startswith("."): continue
        op = t[0]
        if op.startswith("buffer_load_dword") and len(t) > 1:
            d = regs(t[1]); pending |= d; dests |= d
            continue
        is_copy = op.startswith("v_mov_b") or op.startswith("v_accvgpr_write") or op.startswith("scratch_store")
        src_toks = t[1:] if (op.startswith("scratch_store") or op.startswith("buffer_store") or op.startswith("ds_write") or op.startswith("global_")) else t[2:]
        srcs = set()
        for tok in src_toks: srcs |= regs(tok)
        if is_copy:
            if srcs & pending: bad.append((i + 1, txt))
        else:
            pending -= srcs
    print(f"{name}: {len(dests)} request-destination VGPRs, {len(bad)} copies of a register whose request has not been consumed")
    for ln, txt in bad[:40]: print(f"   line {ln}: {txt}")
    bad_total += len(bad)
sys.exit(1 if bad_total else 0)
