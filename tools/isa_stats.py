#!/usr/bin/env python3
"""ISA statistics of one kernel of conv.hip: resource usage, and per straight-line region (between labels / branches)
the MFMA, VALU, LDS, VMEM, scratch and AGPR-move counts.  Usage: tools/isa_stats.py <mangled-name-substring> [conv.s]"""
import collections
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "3d_object_detection_amd", "csrc", "conv.hip")


def main():
    pat = sys.argv[1]
    asm = sys.argv[2] if len(sys.argv) > 2 else "/tmp/conv.s"
    if not os.path.exists(asm) or os.path.getmtime(asm) < os.path.getmtime(SRC):
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=off", "-fno-slp-vectorize",
                               "-munsafe-fp-atomics", "-S", "--cuda-device-only", SRC, "-o", asm] + os.environ.get("PP_EXTRA_FLAGS", "").split(),
                              stderr=subprocess.DEVNULL)
    s = open(asm).read()
    names = [l.split(":")[0] for l in s.splitlines() if l.startswith("_Z") and ":" in l and pat in l.split(":")[0]]
    for name in names:
        i = s.index("\n" + name + ":")
        j = s.index(".Lfunc_end", i)
        body = [l.strip() for l in s[i:j].splitlines()]
        body = [l for l in body if l and not l.startswith(";")]
        meta = s[j:j + 6000]
        info = {k: meta.split(k)[1].split("\n")[0].strip() for k in ("; NumVgprs:", "; NumAgprs:", "; ScratchSize:", "; Occupancy:", "; SGPRBlocks:") if k in meta}
        print("==", name, info)
        reg, regs = collections.Counter(), []
        for l in body:
            op = l.split()[0]
            if l.endswith(":") or op.startswith("s_cbranch") or op == "s_branch":
                if sum(reg.values()) > 12:
                    regs.append((l[:24], reg))
                reg = collections.Counter()
                continue
            k = ("mfma" if op.startswith("v_mfma") else "scratch" if op.startswith("scratch_") else "accmov" if op.startswith("v_accvgpr") else
                 "pk" if op.startswith("v_pk_") else "valu" if op.startswith("v_") else "lds" if op.startswith("ds_") else
                 "vmem" if op.startswith(("buffer_", "global_")) else "smem" if op.startswith("s_load") else "wait" if op.startswith(("s_waitcnt", "s_nop")) else "salu")
            reg[k] += 1
        for end, r in regs:
            print(f"  {sum(r.values()):5d} instr  " + "  ".join(f"{k}={v}" for k, v in sorted(r.items())) + f"   -> {end}")


if __name__ == "__main__":
    main()
