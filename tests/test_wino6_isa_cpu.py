"""ISA audit of wino6_mfma (no GPU needed: hipcc cross-compiles): the kernel issues its vector-memory requests from asm and waits with its
own counts, so a compiler-inserted copy or spill of a destination register between a request and its use takes stale data (round 4 shipped
such a build for a day: one frame of one pass in thirty was off by a few per cent).  tools/inflight_check.py lists such instructions."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(not os.path.exists("/opt/rocm/bin/hipcc"), reason="hipcc not installed")
def test_no_copy_of_an_inflight_register():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "inflight_check.py")], capture_output=True, text=True, timeout=900)
    sys.stdout.write(r.stdout)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert r.stdout.count("wino6_mfma") == 6 and "request-destination VGPRs, 0 copies" in r.stdout
