"""GPU tool: device time of one eager training step per C-ABI entry point and, for the 1x1-conv contractions, per (M, K, N)
shape (`_lib.profile`: HIP events around every call).
    python tools/prof_train_calls.py [workload] [B]"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from tdeed_amd import synth, state_layout, ops, _lib
from tdeed_amd.trainer import TrainEngine
from tdeed_amd.regnet_spec import regnet_spec

wname = sys.argv[1] if len(sys.argv) > 1 else "rny008_b16"
wl = bench.CONFIGS[wname]
cfg, H, W = wl["cfg"], wl["H"], wl["W"]
B = int(sys.argv[2]) if len(sys.argv) > 2 else wl["B"]
T = cfg["clip_len"]
sd = {k: torch.from_numpy(v) for k, v in synth.make_state(state_layout.model_state_shapes(cfg), 0).items()}
eng = TrainEngine(cfg, sd, torch.bfloat16, "cuda", lr=1e-4)
frames = ops.fill_u8_hash((B, T, 3, H, W), 1000, "cuda")
lab_np, labD_np = synth.labels(5, B, T, cfg["num_classes"], max(cfg["radi_displacement"], 1))
lab = torch.from_numpy(lab_np).cuda()
labD = torch.from_numpy(labD_np).float().cuda() if cfg["radi_displacement"] else None
C = regnet_spec(cfg["feature_arch"]).feat_dim
masks = [((torch.rand((B, T, C), device="cuda") >= 0.5).to(torch.bfloat16) * 2.0) for _ in range(2 if cfg["radi_displacement"] else 1)]
for _ in range(2):
    eng.step(frames, lab, labD, drop_masks=masks)
with _lib.profile() as pr:
    eng.step(frames, lab, labD, drop_masks=masks)
s = pr.summary()
tot = sum(d["ms"] for d in s.values())
print(f"{wname} B={B}: {tot:.2f} ms of device time inside C-ABI calls (one eager step)")
for k, d in sorted(s.items(), key=lambda kv: -kv[1]["ms"]):
    print(f"  {k:34s} {d['ms']:8.3f} ms {d['calls']:5d} calls {d['ms'] / tot * 100:5.1f} %")
sc = pr.by_scope()
print("per scope (stage / block, forward and backward), ms of device time and the three largest entry points:")
for k, d in sc.items():
    tt = sum(d.values())
    top = sorted(d.items(), key=lambda kv: -kv[1])[:(99 if os.environ.get("PROF_ALL") else 3)]
    print(f"  {k or '-':16s} {tt:8.3f} ms   " + "  ".join(f"{n.replace('tdeed_', '')} {v:.2f}" for n, v in top))
for entry, idx in (("tdeed_gemm_fwd", (7, 8, 9)), ("tdeed_wgrad", None)):
    g = s.get(entry)
    if g is None or idx is None:
        continue
    by = {}
    # re-time per call: the summary keeps args per call in order; events are gone, so re-run under a fresh profile
    with _lib.profile() as pr2:
        eng.step(frames, lab, labD, drop_masks=masks)
    torch.cuda.synchronize()
    for name, a, b, args, *_ in pr2.rec:
        if name != entry:
            continue
        key = tuple(args[i] for i in idx) + (bool(args[14]), bool(args[25]))      # residual, colpart
        d = by.setdefault(key, [0.0, 0])
        d[0] += a.elapsed_time(b)
        d[1] += 1
    print(f"{entry} by (M, K, N, residual, colpart):")
    for key, (ms, n) in sorted(by.items(), key=lambda kv: -kv[1][0]):
        M, K, N = key[:3]
        gb = (M * K + N * K + M * N * (2 if key[3] else 1)) * 2 / 1e9
        print(f"  M={M:8d} K={K:4d} N={N:4d} res={int(key[3])} stats={int(key[4])}: {ms:7.3f} ms / {n:2d} calls = {ms / n * 1e3:7.1f} us"
              f"  {gb * n / ms:7.1f} GB/s  {2.0 * M * K * N * n / ms / 1e9:6.1f} TF")
