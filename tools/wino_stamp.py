"""Where a Winograd wave spends its cycles (diagnostic build with -DPP_WINO_STAMP=1, see conv.hip).
Runs the backbone on a few frames and prints the s_memtime shares of the chunk-loop segments (wave 0 of every
workgroup).  The stamps' fences forbid overlaps the real kernel has: read the SHARES, not the lengths."""
import ctypes, importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("PP_HIP_LIB", os.path.join(ROOT, "3d_object_detection_amd", "csrc", "_build", "libpp_stamp.so"))
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
buf = [int(v) for v in dbuf.cpu().numpy()]
pre, steps, bar, epi, nch, nt = [buf[i] for i in range(6)]
e1, pro = buf[6], buf[7]
tot = pre + steps + bar + epi + pro
print(f"per tile: prologue {pro / nt:.0f}  epilogue up to stores {e1 / nt:.0f}  statistics {(epi - e1) / nt:.0f}  (chunks per tile {nch / nt:.1f})")
print(f"chunks {nch}  tiles {nt}  cycles/chunk: pre-steps {pre / nch:.0f}  steps {steps / nch:.0f}  barrier {bar / nch:.0f}   epilogue+setup per tile {epi / max(nt, 1):.0f}")
print(f"shares: pre-steps {pre / tot:.3f}  steps {steps / tot:.3f}  barrier {bar / tot:.3f}  epilogue {epi / tot:.3f}  tile prologue {pro / tot:.3f}")
