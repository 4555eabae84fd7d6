"""Where a wino4_mfma wave spends its cycles (diagnostic build: make -C 3d_object_detection_amd/csrc stamp).
Forces the given wino4 tiling on every stride-1 3x3 layer, runs the backbone on a few frames and prints the s_memtime
sums of the chunk-loop segments (lane 0 of wave 0 of every workgroup).  The stamps fence the segments (lgkmcnt(0) +
sched_barrier): read shares, not absolute lengths.  Usage: tools/wino4_stamp.py [frames] [tiling substring]"""
import ctypes, importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("PP_HIP_LIB", os.path.join(ROOT, "3d_object_detection_amd", "csrc", "_build", "libpp_stamp.so"))
os.environ["PP_FORCE_VARIANT"] = sys.argv[2] if len(sys.argv) > 2 else "wino4 tw4 bx2"
synth = importlib.import_module("3d_object_detection_amd.synth")
eng_mod = importlib.import_module("3d_object_detection_amd.engine")
_lib = importlib.import_module("3d_object_detection_amd._lib")
cfg = synth.load_config("eight_20cm")
cfg["device"] = torch.device("cuda:0")
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 8
eng = eng_mod.Engine(dict(cfg), device_index=0, max_batch=nb)
eng.load_state_dict(synth.seeded_state_dict(0))
clouds = [torch.from_numpy(synth.lidar_cloud("eight_20cm", seed=1000 + i)).cuda() for i in range(nb)]
lib = _lib.load()
dbuf = torch.zeros(8, dtype=torch.int64, device='cuda')
eng.infer_batch(clouds)
torch.cuda.synchronize()
lib.pp_debug_set_stamp_buffer(ctypes.c_void_p(dbuf.data_ptr()))
for _ in range(3):
    eng.infer_batch(clouds)
torch.cuda.synchronize()
pre, steps, bar, epi, nch, nt, p1, p2 = [int(v) for v in dbuf.cpu().numpy()][:8]
tot = pre + steps + bar + epi
print(f"{os.environ['PP_FORCE_VARIANT']}: chunks {nch} tiles {nt} ({nch / nt:.1f} chunks per tile)")
print(f"cycles per chunk: top (wait for loads, normalise, advance) {pre / nch:.0f}   32 steps {steps / nch:.0f} (matrix pipe floor 4096)   barrier {bar / nch:.0f}")
print(f"cycles per tile: epilogue {epi / nt:.0f} (output transform {p1 / nt:.0f}, residual add + stores + statistics {p2 / nt:.0f}, rest {(epi - p1 - p2) / nt:.0f})   whole tile {tot / nt:.0f}")
print(f"shares: top {pre / tot:.3f}  steps {steps / tot:.3f}  barrier {bar / tot:.3f}  epilogue {epi / tot:.3f}")
