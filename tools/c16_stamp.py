"""Where a conv16 workgroup spends its cycles (diagnostic build: make -C 3d_object_detection_amd/csrc stamp16).
Forces the given conv16 tiling on every 3x3 layer it fits (others keep the tuner's pick), runs the fp16 backbone on a few frames and
prints the s_memtime sums of an item's phases (lane 0 of wave 0 of every workgroup, all conv16 launches of the pass together --
narrow with PP_STAMP_CIN=<input channels>).  Usage: tools/c16_stamp.py [frames] [tiling substring] [mode]"""
import ctypes, importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("PP_HIP_LIB", os.path.join(ROOT, "3d_object_detection_amd", "csrc", "_build", "libpp_stamp16.so"))
os.environ["PP_FORCE_VARIANT"] = sys.argv[2] if len(sys.argv) > 2 else "c16 s1"
mode = sys.argv[3] if len(sys.argv) > 3 else "fp16"
synth = importlib.import_module("3d_object_detection_amd.synth")
eng_mod = importlib.import_module("3d_object_detection_amd.engine")
_lib = importlib.import_module("3d_object_detection_amd._lib")
cfg = synth.load_config("eight_20cm")
cfg["device"] = torch.device("cuda:0")
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 16
eng = eng_mod.Engine(dict(cfg), device_index=0, max_batch=nb, precision=mode)
eng.load_state_dict(synth.seeded_state_dict(0))
clouds = [torch.from_numpy(synth.lidar_cloud("eight_20cm", seed=1000 + i)).cuda() for i in range(nb)]
lib = _lib.load()
dbuf = torch.zeros(8, dtype=torch.int64, device="cuda")
eng.infer_batch(clouds)
torch.cuda.synchronize()
lib.pp_debug_set_stamp_buffer(ctypes.c_void_p(dbuf.data_ptr()))
for _ in range(3):
    eng.infer_batch(clouds)
torch.cuda.synchronize()
pro, iss, taps, bar1, commit, bar2, epi, items = [int(v) for v in dbuf.cpu().numpy()][:8]
tot = pro + iss + taps + bar1 + commit + bar2 + epi
print(f"{os.environ['PP_FORCE_VARIANT']} ({mode}, PP_STAMP_CIN={os.environ.get('PP_STAMP_CIN')}): {items} items, {tot / items:.0f} cycles per item")
for name, v in (("prologue (first step staged, exposed)", pro), ("loads issued", iss), ("tap loop (MFMAs + operand reads)", taps), ("barrier after the taps", bar1),
                ("commit + rest of staging", commit), ("second barrier", bar2), ("epilogue + item setup", epi)):
    print(f"  {name:42s} {v / items:9.0f} cycles per item  {v / tot:6.3f}")
