"""Headline benchmark: clips/s of the T-DEED forward (BASELINE.json configs[1]: RegNetY-200MF + GSF + SGP,
L=100, 224x224, batch 8 per GPU, bf16 inference) on N MI355X, one process per GPU.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A step = one pass of the whole hot path (uint8 clips resident in HBM -> per-frame logits) over one batch
of B synthetic clips per GPU.  Clips shard over ranks with no data-path collective (inference), so
scaling is weak: value = N * B * K / max-over-ranks(time).  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import tdeed_amd  # noqa: E402,F401
from tdeed_amd import synth, state_layout, ops, dist as tdist  # noqa: E402
from tdeed_amd.engine import ForwardEngine  # noqa: E402

CONFIGS = {
    # BASELINE.json configs[1] -- the configuration the metric is quoted on
    "rny002_b8": dict(cfg=dict(feature_arch="rny002_gsf", clip_len=100, crop_dim=224, n_layers=2, sgp_ks=7, sgp_r=4,
                               num_classes=4, radi_displacement=2), B=8, H=224, W=224),
    "rny008_b16": dict(cfg=dict(feature_arch="rny008_gsf", clip_len=100, crop_dim=224, n_layers=3, sgp_ks=7, sgp_r=4,
                                num_classes=4, radi_displacement=2), B=16, H=224, W=224),
    "snb_t250_b4": dict(cfg=dict(feature_arch="rny008_gsf", clip_len=250, crop_dim=None, n_layers=2, sgp_ks=9, sgp_r=4,
                                 num_classes=12, radi_displacement=4), B=4, H=224, W=224),
}
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
MFMA_PEAK_TF = {torch.bfloat16: 2500.0, torch.float32: 157.3}


def kernel_profile(eng, plan, reps=3):
    """Per-kernel-family device time, measured live with HIP events on the launch stream (eager replay
    of the same launches the graph holds)."""
    agg = {}
    sgp_ms_acc = 0.0
    st = torch.cuda.current_stream()
    for r in range(reps + 1):
        for _ in range(3):            # keep the GPU busy so the host runs ahead of it: the event
            for s in plan.steps:      # intervals below then contain device time only, no launch gaps
                s.fn()
        evs = []
        for s in plan.steps:
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(st)
            s.fn()
            b.record(st)
            evs.append((s, a, b))
        st.synchronize()
        if r == 0:
            continue                      # first pass warms caches / code objects
        for s, a, b in evs:
            d = agg.setdefault(s.kernel, dict(ms=0.0, launches=0, bytes=0, flops=0))
            d["ms"] += a.elapsed_time(b)
            if s.name.startswith("_temp_fine."):
                sgp_ms_acc += a.elapsed_time(b)
            d["launches"] += 1
            d["bytes"] += s.bytes
            d["flops"] += s.flops
    for d in agg.values():
        for k in ("ms", "launches", "bytes", "flops"):
            d[k] = d[k] / reps
    return agg, sgp_ms_acc / reps


def cpu_baseline():
    """The oracle (our CPU port of the reference forward) timed on this host: BASELINE.json configs[0]
    = FineDiving_small, 1 synthetic clip, fp32, all host cores.  Bounded: 1 warm-up + 3 timed clips."""
    from oracle import tdeed_oracle as O
    from tdeed_amd.regnet_spec import regnet_spec
    c = CONFIGS["rny002_b8"]["cfg"]
    sd = O.as_torch_state(synth.make_state(state_layout.model_state_shapes(c), 0))
    clip = torch.from_numpy(synth.uint8_clip(1000, (1, 100, 3, 224, 224)))
    spec = regnet_spec(c["feature_arch"])
    ts = []
    with torch.no_grad():
        for i in range(4):
            t0 = time.perf_counter()
            O.forward(clip, sd, c, spec)
            ts.append(time.perf_counter() - t0)
    best = sorted(ts[1:])[len(ts[1:]) // 2]
    return dict(value=round(1.0 / best, 4), unit="clips/s", cores=torch.get_num_threads(), kind="port",
                sample="FineDiving_small forward, 1 synthetic clip (100x3x224x224), fp32, median of 3 after 1 warm-up",
                sec_per_clip=round(best, 4))


def _dist_setup():
    """One process per GPU under torch.distributed.run; RCCL (backend "nccl").  TDEED_DIST_BACKEND=gloo lets several
    ranks share one GPU to exercise the multi-process path on a single-GPU box (functional check, not a measurement)."""
    rank, local, world = tdist.env_world()
    backend = os.environ.get("TDEED_DIST_BACKEND", "nccl")
    local_dev = local % max(torch.cuda.device_count(), 1) if backend != "nccl" else local
    torch.cuda.set_device(local_dev)
    tdist.init(backend=backend, device=torch.device("cuda", local_dev))   # no-op for a single process
    return rank, local, world, f"cuda:{local_dev}"


def main_train(a):
    """BASELINE.json configs[2]/[3]-style measurement: optimisation steps per second of the training path.
    A step = train-mode forward + CE/MSE loss + full backward + (N > 1: one RCCL all-reduce of the flat fp32 gradient
    buffer) + fused AdamW, on synthetic uint8 clips resident in HBM, dropout masks fixed."""
    from tdeed_amd.trainer import TrainEngine
    from tdeed_amd.regnet_spec import regnet_spec
    rank, local, world, dev = _dist_setup()
    wl = CONFIGS[a.workload]
    cfg, B, H, W = wl["cfg"], wl["B"], wl["H"], wl["W"]
    dt = torch.bfloat16 if a.dtype == "bf16" else torch.float32
    T = cfg["clip_len"]
    sd = {k: torch.from_numpy(v) for k, v in synth.make_state(state_layout.model_state_shapes(cfg), 0).items()}
    eng = TrainEngine(cfg, sd, dt, dev, lr=1e-4)
    frames = ops.fill_u8_hash((B, T, 3, H, W), 1000 + rank, dev)
    lab_np, labD_np = synth.labels(5 + rank, B, T, cfg["num_classes"], max(cfg["radi_displacement"], 1))
    lab = torch.from_numpy(lab_np).to(dev)
    labD = torch.from_numpy(labD_np).float().to(dev) if cfg["radi_displacement"] else None
    C = regnet_spec(cfg["feature_arch"]).feat_dim
    masks = [((torch.rand((B, T, C), device=dev) >= 0.5).to(dt) * 2.0) for _ in range(2 if cfg["radi_displacement"] else 1)]
    ar = tdist.all_reduce_mean_ if world > 1 else None
    if a.no_graph:
        step = lambda: eng.step(frames, lab, labD, drop_masks=masks, all_reduce=ar)                      # noqa: E731
    else:
        hnd = eng.build_graph(B, H, W)
        step = lambda: eng.step_graph(hnd, frames, lab, labD, drop_masks=masks, all_reduce=ar)           # noqa: E731
    for _ in range(max(a.warmup, 1)):
        step()
    torch.cuda.synchronize()
    tdist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        loss = step()
    torch.cuda.synchronize()
    el = tdist.max_over_ranks(time.perf_counter() - t0, device=dev)
    tdist.barrier()
    if rank == 0:
        out = dict(metric="clips/sec (L=%d, %dx%d, %s) training step (fwd + loss + bwd + AdamW)" % (T, H, W, a.dtype),
                   value=round(world * B * a.steps / el, 2), unit="clips/s", n_gpus=world, steps=a.steps, warmup=a.warmup,
                   ms_per_step=round(el / a.steps * 1e3, 3), higher_is_better=True, scaling="weak", vs_baseline=None,
                   dtype=a.dtype, data="synthetic",
                   config=dict(workload=f"{a.workload}: {cfg['feature_arch']} + ed_sgp_mixer n_layers={cfg['n_layers']}, L={T}, "
                                        f"{H}x{W}, batch {B}/GPU, training step, random-init weights", clips_per_gpu=B,
                               parallelism=f"dp{world}" + (" (RCCL all-reduce of one flat fp32 gradient buffer)" if world > 1 else ""),
                               grad_buffer_mb=round(eng.params.numel * 4 / 2 ** 20, 1), final_loss=round(float(loss[0]), 4),
                               hip_graph=not a.no_graph),
                   roofline=None, cpu_baseline=None)
        print(json.dumps(out))
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="rny002_b8", choices=list(CONFIGS))
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--inflight", type=int, default=2,
                    help="batches in flight: consecutive steps alternate between this many independent buffer sets / HIP "
                         "graphs on their own streams, so the latency-bound tail of one batch overlaps the next one's "
                         "head (every step still runs the full forward on its own batch of B clips)")
    ap.add_argument("--mode", default="infer", choices=["infer", "train"],
                    help="infer (default, the BASELINE metric): forward of configs[1]; train: one optimisation step "
                         "(train-mode forward + loss + backward + fused AdamW, gradient all-reduce over RCCL when N > 1)")
    a = ap.parse_args()
    if a.mode == "train":
        return main_train(a)

    rank, local, world, dev = _dist_setup()
    wl = CONFIGS[a.workload]
    cfg, B, H, W = wl["cfg"], wl["B"], wl["H"], wl["W"]
    dt = torch.bfloat16 if a.dtype == "bf16" else torch.float32
    sd = synth.make_state(state_layout.model_state_shapes(cfg), 0)       # random-init weights of the architecture
    depth = max(1, a.inflight)
    streams = [torch.cuda.Stream() for _ in range(depth)]
    stream = streams[0]
    with torch.cuda.stream(stream):
        eng = ForwardEngine(cfg, sd, dt, dev, use_graph=not a.no_graph)
        T = cfg["clip_len"]
        plans = [eng.plan(B, H, W, slot=i) for i in range(depth)]
        plan = plans[0]
        # synthetic uint8 clips, generated on the device straight into each plan's input buffer
        for i, pl in enumerate(plans):
            eng.set_frames(pl, ops.fill_u8_hash((B, T, 3, H, W), 1000 + rank + 97 * i, dev))
    torch.cuda.synchronize()

    def run(n):
        for i in range(n):
            with torch.cuda.stream(streams[i % depth]):
                eng.run_plan(plans[i % depth])

    run(max(a.warmup, depth))
    torch.cuda.synchronize()
    tdist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(a.steps)
    torch.cuda.synchronize()
    el = tdist.max_over_ranks(time.perf_counter() - t0, device=dev)
    tdist.barrier()
    with torch.cuda.stream(stream):
        prof, sgp_stage_ms = kernel_profile(eng, plan) if rank == 0 else (None, None)

    if rank == 0:
        ms = el / a.steps * 1e3
        value = world * B * a.steps / el
        dom = max(prof.items(), key=lambda kv: kv[1]["ms"])
        name, d = dom
        per_launch_s = d["ms"] / max(d["launches"], 1) * 1e-3
        gbs = d["bytes"] / max(d["launches"], 1) / per_launch_s / 1e9
        tfs = d["flops"] / max(d["launches"], 1) / per_launch_s / 1e12
        # bound: whichever roof the kernel's algorithmic intensity puts it under
        mfma_bound = name == "gemm" and (d["flops"] / max(d["bytes"], 1)) > (MFMA_PEAK_TF[dt] * 1e12 / (HBM_PEAK_GBS * 1e9))
        roof = dict(kernel=name, bound="mfma" if mfma_bound else "hbm",
                    achieved=round(tfs if mfma_bound else gbs, 2), peak=MFMA_PEAK_TF[dt] if mfma_bound else HBM_PEAK_GBS,
                    unit="TFLOP/s" if mfma_bound else "GB/s",
                    frac=round((tfs / MFMA_PEAK_TF[dt]) if mfma_bound else (gbs / HBM_PEAK_GBS), 4), traffic=None,
                    launches_per_step=d["launches"], ms_per_step=round(d["ms"], 4),
                    share_of_step=round(d["ms"] / sum(x["ms"] for x in prof.values()), 3))
        # HBM bytes per launch of that kernel family from the PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE run
        # separately on this same command, corrected as MI355X_MICROARCH.md prescribes; tools/summarize_pmc.py)
        try:
            with open(os.path.join(ROOT, "profiles", "r01_hbm_traffic.json")) as fh:
                tr = json.load(fh)["kernels"].get(name)
            if tr is not None and a.workload == "rny002_b8" and a.dtype == "bf16":
                roof["traffic"] = tr["hbm_bytes_per_launch"]
                roof["algorithmic_bytes_per_launch"] = int(d["bytes"] / max(d["launches"], 1))
        except (OSError, KeyError, ValueError):
            pass
        sgp_steps = [s for s in plan.steps if s.name.startswith("_temp_fine.")]
        # SGP encoder-decoder against the HBM roof with the ALGORITHMIC byte count of SURVEY.md section 8d:
        # es * [B*C*Sigma_T + W]: every block/mixer reads its inputs once and writes its output once, weights once
        from tdeed_amd.regnet_spec import pyramid_lengths, sgp_up_size
        n_l, Cc = cfg["n_layers"], eng.pw.spec.feat_dim
        lens = pyramid_lengths(T, n_l)
        sig = sum(2 * lens[i] + lens[i + 1] for i in range(n_l)) + 2 * lens[n_l] \
            + sum((2 * lens[l] + lens[l + 1]) + 2 * lens[l] for l in range(n_l))
        ks_, up_ = cfg["sgp_ks"], sgp_up_size(cfg["sgp_ks"], cfg["sgp_r"])
        Wsgp = (2 * n_l + 1) * (8 * Cc * Cc + (2 * ks_ + up_ + 16) * Cc) + n_l * (14 * Cc * Cc + (4 * ks_ + 2 * up_ + 26) * Cc)
        es_ = 2 if dt == torch.bfloat16 else 4
        sgp_bytes = es_ * (B * Cc * sig + Wsgp)
        # (stage time = all launches named _temp_fine.*, timed by step name: its contractions share the "gemm"
        # family with the s4 trunk layers)
        kernels = {k: dict(ms=round(v["ms"], 4), launches=v["launches"],
                           GBps=round(v["bytes"] / (v["ms"] * 1e-3) / 1e9, 1) if v["ms"] > 0 else 0,
                           TFLOPs=round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 2) if v["ms"] > 0 else 0)
                   for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["ms"])}
        out = dict(metric="clips/sec (L=100, 224^2, bf16) forward, per-frame logits", value=round(value, 2),
                   unit="clips/s", n_gpus=world, steps=a.steps, warmup=a.warmup, ms_per_step=round(ms, 4),
                   higher_is_better=True, scaling="weak", vs_baseline=None, dtype=a.dtype, data="synthetic",
                   config=dict(workload=f"{a.workload}: {cfg['feature_arch']} + ed_sgp_mixer n_layers={cfg['n_layers']} "
                                        f"ks={cfg['sgp_ks']}, L={T}, {H}x{W}, batch {B}/GPU, inference forward, "
                                        "random-init weights", clips_per_gpu=B, parallelism=f"dp{world} (clip-sharded, no collective)",
                               hip_graph=not a.no_graph, batches_in_flight=depth),
                   roofline=roof, kernels=kernels,
                   roofline_sgp=dict(bound="hbm", algorithmic_bytes=int(sgp_bytes), sigma_T=int(sig), weights=int(Wsgp),
                                     ms=round(sgp_stage_ms, 4), achieved=round(sgp_bytes / (sgp_stage_ms * 1e-3) / 1e9, 2),
                                     peak=HBM_PEAK_GBS, unit="GB/s",
                                     frac=round(sgp_bytes / (sgp_stage_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                     launches=len(sgp_steps)),
                   cpu_baseline=None)
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out))
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
