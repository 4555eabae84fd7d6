"""GPU probe: reduced-precision modes -- tilings, logit deviation from the fp32 path, detections, frames/s.  python tools/prec_probe.py [config] [batch] [modes]"""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
synth = importlib.import_module("3d_object_detection_amd.synth")
eng_mod = importlib.import_module("3d_object_detection_amd.engine")
name = sys.argv[1] if len(sys.argv) > 1 else "nuscene"
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 8
modes = sys.argv[3].split(",") if len(sys.argv) > 3 else ["fp32", "bf16x3", "bf16", "fp16"]
cfg = synth.load_config(name); cfg["device"] = torch.device("cuda:0")
sd = synth.seeded_state_dict(1, cls_bias=-3.0)
clouds = [torch.from_numpy(synth.lidar_cloud(name, seed=77 + i)).cuda() for i in range(nb)]
ref = None
for mode in modes:
    t0 = time.time()
    eng = eng_mod.Engine(dict(cfg), max_batch=nb, precision=mode)
    eng.load_state_dict(sd)
    torch.cuda.synchronize()
    t_build = time.time() - t0
    det, cnt = eng.infer_batch(clouds)
    torch.cuda.synchronize()
    lg = {k: eng.fetch(0, k).cpu().numpy() for k in ("cls", "box", "dir")}
    rpn = eng.fetch(0, "rpn").cpu().numpy()
    if ref is None:
        ref = (lg, rpn)
    dev = {k: float(np.abs(lg[k] - ref[0][k]).max()) for k in lg}
    drpn = float(np.abs(rpn - ref[1]).max())
    t0 = time.time()
    for _ in range(5):
        eng.infer_batch(clouds, det, cnt)
    torch.cuda.synchronize()
    dt = (time.time() - t0) / 5
    til = [t["tiling"] for t in eng.layer_tilings()]
    print(f"[{mode}] build {t_build:.1f}s  {nb / dt:.1f} frames/s ({dt / nb * 1e3:.3f} ms/frame)  logit dev vs fp32 {dev}  rpn dev {drpn:.3e}  det {cnt[:, 0].tolist()[:4]}", flush=True)
    print("   ", " | ".join(til), flush=True)
    del eng
