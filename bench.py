"""Headline benchmark: clips/s of the T-DEED forward (BASELINE.json configs[1]: RegNetY-200MF + GSF + SGP,
L=100, 224x224, batch 8 per GPU, bf16 inference) on N MI355X, one process per GPU.

    python bench.py --gpus 1 --steps 50 --warmup 10
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W
    python bench.py --mode train --workload rny008_b16          # BASELINE configs[2]: one optimisation step

A step = one pass of the whole hot path (uint8 clips resident in HBM -> per-frame logits) over one batch
of B synthetic clips per GPU.  Clips shard over ranks with no data-path collective (inference), so
scaling is weak: value = N * B * K / max-over-ranks(time).  W untimed warm-up steps, then `--repeats` timed regions of
EXACTLY K steps each, every one bracketed by barrier + device synchronisation on both sides; the line reports the
MEDIAN region (all regions are listed in `ms_per_step_repeats`).  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import statistics
import sys
import time
from types import SimpleNamespace

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def _requested_gpus(argv):
    """--gpus N as given on the command line (1 if absent), read without argparse so that it is known before torch is
    imported and before anything touches the GPU."""
    n = 1
    for i, tok in enumerate(argv):
        if tok == "--gpus" and i + 1 < len(argv):
            n = int(argv[i + 1])
        elif tok.startswith("--gpus="):
            n = int(tok.split("=", 1)[1])
    return n


def _self_launch(n, argv):
    """`python bench.py --gpus N` with no torchrun environment around it: start N CHILD ranks of this same script (one
    process per GPU, RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, rendezvous on 127.0.0.1), relay rank 0's JSON line and
    leave with the worst child's return code.  The parent never imports torch and never initialises the GPU; nothing is
    re-exec'd (a child is a fresh interpreter started before any HIP call of its own)."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0",
                   TDEED_BENCH_SELF_LAUNCHED="1")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env, cwd=os.getcwd(),
                                      stdout=(subprocess.PIPE if r == 0 else subprocess.DEVNULL), text=True))
    import threading
    got = []
    rd = threading.Thread(target=lambda: got.append(procs[0].stdout.read()), daemon=True)
    rd.start()
    # a rank that dies before the rendezvous would leave the others waiting for it: end them (exact PIDs) and report
    while any(p.poll() is None for p in procs):
        if any(p.poll() not in (None, 0) for p in procs):
            for p in procs:
                if p.poll() is None:
                    p.terminate()
        time.sleep(0.2)
    rd.join(timeout=10)
    sys.stdout.write("".join(got))
    sys.stdout.flush()
    bad = [p.returncode for p in procs if p.returncode != 0]
    return (bad[0] if bad else 0)


if __name__ == "__main__":
    _n = _requested_gpus(sys.argv[1:])
    _w = os.environ.get("WORLD_SIZE")
    if _w is None and _n > 1:
        sys.exit(_self_launch(_n, sys.argv[1:]))
    if _w is not None and int(_w) != _n:
        sys.stderr.write(f"bench.py: --gpus {_n} disagrees with WORLD_SIZE={_w} of the launching environment\n")
        sys.exit(2)

import torch  # noqa: E402
import tdeed_amd  # noqa: E402,F401
from tdeed_amd import synth, state_layout, ops, dist as tdist  # noqa: E402
from tdeed_amd.engine import ForwardEngine  # noqa: E402

CONFIGS = {
    # BASELINE.json configs[1] -- the configuration the metric is quoted on
    "rny002_b8": dict(cfg=dict(feature_arch="rny002_gsf", clip_len=100, crop_dim=224, n_layers=2, sgp_ks=7, sgp_r=4,
                               num_classes=4, radi_displacement=2), B=8, H=224, W=224),
    # configs[2] (train step) / configs[3] per-GPU share is rny008 at B=8
    "rny008_b16": dict(cfg=dict(feature_arch="rny008_gsf", clip_len=100, crop_dim=224, n_layers=3, sgp_ks=7, sgp_r=4,
                                num_classes=4, radi_displacement=2), B=16, H=224, W=224),
    "rny008_b8": dict(cfg=dict(feature_arch="rny008_gsf", clip_len=100, crop_dim=224, n_layers=3, sgp_ks=7, sgp_r=4,
                               num_classes=4, radi_displacement=2), B=8, H=224, W=224),
    # configs[4] per-GPU share: SoccerNetBall hyper-parameters at T=250
    "snb_t250_b4": dict(cfg=dict(feature_arch="rny008_gsf", clip_len=250, crop_dim=None, n_layers=2, sgp_ks=9, sgp_r=4,
                                 num_classes=12, radi_displacement=4), B=4, H=224, W=224),
}
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
MFMA_PEAK_TF = {torch.bfloat16: 2500.0, torch.float32: 157.3}
TRAFFIC_FILE = os.path.join("profiles", "r06_hbm_traffic.json")
TRAFFIC_FILE_800MF = os.path.join("profiles", "r06_hbm_traffic_800mf_b16.json")
TRAFFIC_FILE_SNB = os.path.join("profiles", "r06_hbm_traffic_snb_t250_b4.json")
TRAIN_TRAFFIC_FILE = os.path.join("profiles", "r06_train_hbm_traffic.json")
# C-ABI entry -> kernel family of tools/summarize_pmc.py (what the counter passes are keyed by)
DIST_INFO = dict(backend="none", ranks=1)      # _dist_setup(): what the process group itself counted
TRAIN_FAMILY = {"tdeed_gemm_fwd": "gemm", "tdeed_bn_train_bwd": "bn_bwd", "tdeed_wgrad": "wgrad",
                "tdeed_narrow_conv1_bwd": "narrow_conv_bwd"}


def _masked_streams(depth, mode):
    """`depth` HIP streams, each confined to its own 1/depth of the CUs: every `depth`-th CU ("interleave") or a contiguous
    block of the CU numbering ("block").  An experiment: kernels of different batches then never queue for the
    same CUs."""
    import ctypes
    hip = ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
    ncu = torch.cuda.get_device_properties(0).multi_processor_count
    words = (ncu + 31) // 32
    out = []
    for s_ in range(depth):
        bits = [0] * words
        for i in range(ncu):
            own = (i % depth == s_) if mode == "interleave" else (i * depth // ncu == s_)
            if own:
                bits[i // 32] |= 1 << (i % 32)
        arr = (ctypes.c_uint32 * words)(*bits)
        h = ctypes.c_void_p()
        rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(h), ctypes.c_uint32(words), arr)
        if rc != 0:
            raise RuntimeError(f"hipExtStreamCreateWithCUMask failed: {rc}")
        out.append(torch.cuda.ExternalStream(h.value))
    return out


def git_head():
    from tdeed_amd import buildinfo
    return buildinfo.head()


def timed_regions(run, steps, repeats, dev, streams):
    """`repeats` regions of exactly `steps` steps.  Per region: barrier + synchronize, host clock, HIP events on the
    launching streams (start on every stream before the first step, end after the last), synchronize + barrier.
    Returns (wall seconds per region [max over ranks], event ms per region [this rank])."""
    walls, evs = [], []
    for _ in range(repeats):
        torch.cuda.synchronize()
        tdist.barrier()
        torch.cuda.synchronize()
        starts = [torch.cuda.Event(enable_timing=True) for _ in streams]
        ends = [torch.cuda.Event(enable_timing=True) for _ in streams]
        t0 = time.perf_counter()
        for s, e in zip(streams, starts):
            e.record(s)
        run(steps)
        for s, e in zip(streams, ends):
            e.record(s)
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        walls.append(tdist.max_over_ranks(el, device=dev))
        tdist.barrier()
        # device time of the region: first start -> last end
        t_end = max(starts[0].elapsed_time(e) for e in ends)
        t_beg = min(starts[0].elapsed_time(s) for s in starts)
        evs.append(t_end - t_beg)
    return walls, evs


def check_timed_outputs(eng, plans, B, T, H, W, rank, dev, heads=None):
    """What the timed plans left in their output buffers: every slot's logits finite, different slots (different clips)
    different, and slot 0 equal to a fresh forward of the same clips through a plan of its own (bf16: <= 2e-2; the same
    kernels in the same order -- in practice bit-identical).  Raises instead of printing a line whose logits are garbage."""
    torch.cuda.synchronize()
    if os.environ.get("TDEED_BENCH_NOCHECK") == "1":             # timing experiments that leave the outputs incomplete on purpose
        return dict(skipped=True)
    if heads is None:
        heads = [p.head_out.float().clone() for p in plans]
    for i, h in enumerate(heads):
        if not bool(torch.isfinite(h).all()):
            raise RuntimeError(f"bench: non-finite logits in the timed plan of slot {i}")
    for i in range(1, len(heads)):
        if float((heads[i] - heads[0]).abs().max()) < 1e-3:
            raise RuntimeError(f"bench: slots 0 and {i} hold the same logits (buffer aliasing between in-flight batches?)")
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        ref_plan = eng.plan(B, H, W, slot=len(plans))                       # a buffer set none of the timed plans touches
        eng.set_frames(ref_plan, ops.fill_u8_hash((B, T, 3, H, W), 1000 + rank, dev))
        eng.run_plan(ref_plan)
        st.synchronize()
        err = float((ref_plan.head_out.float() - heads[0]).abs().max())
    if err > 2e-2:
        raise RuntimeError(f"bench: slot 0's logits differ from a fresh forward of the same clips by {err}")
    return dict(slots=len(plans), finite=True, slots_differ=True, slot0_vs_fresh_plan_max_abs=round(err, 6))


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


def sgp_stage_time(plan, reps=30):
    """Device time of the SGP encoder-decoder alone: the launches named _temp_fine.* of each chain (one chain behind the
    sub-batch join, or one per sub-batch) captured into a HIP graph of their own and replayed back to back on one
    stream -- device time including the kernel-to-kernel boundaries, without host launch gaps."""
    import ctypes
    from tdeed_amd import _lib
    st = torch.cuda.current_stream()
    out = []
    chains = [plan.tail.steps] if getattr(plan, "tail", None) is not None else [sub.steps for sub in plan.subs]
    for chain in chains:
        steps = [s for s in chain if s.name.startswith("_temp_fine.")]
        for s in steps:
            s.fn()
        st.synchronize()
        h = ctypes.c_void_p()
        _lib.call("tdeed_graph_begin", st.cuda_stream)
        try:
            for s in steps:
                s.fn()
        finally:
            _lib.call("tdeed_graph_end", st.cuda_stream, ctypes.byref(h))
        for _ in range(3):
            _lib.call("tdeed_graph_launch", h, st.cuda_stream)
        st.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(st)
        for _ in range(reps):
            _lib.call("tdeed_graph_launch", h, st.cuda_stream)
        b.record(st)
        st.synchronize()
        out.append((a.elapsed_time(b) / reps, len(steps)))
        _lib.call("tdeed_graph_destroy", h)
    return out


def dominant_roofline(prof, dt, traffic_file):
    """`roofline` of the kernel family with the largest share of the serial kernel time (kernel_profile): algorithmic bytes
    (or flops) per launch over the live-measured average launch duration; `traffic` only from a committed counter file."""
    name, d = max(prof.items(), key=lambda kv: kv[1]["ms"])
    per_launch_s = d["ms"] / max(d["launches"], 1) * 1e-3
    gbs = d["bytes"] / max(d["launches"], 1) / per_launch_s / 1e9
    tfs = d["flops"] / max(d["launches"], 1) / per_launch_s / 1e12
    # bound: whichever roof the kernel's algorithmic intensity puts it under
    mfma_bound = name.startswith(("gemm", "bneck")) and (d["flops"] / max(d["bytes"], 1)) > (MFMA_PEAK_TF[dt] * 1e12 / (HBM_PEAK_GBS * 1e9))
    roof = dict(kernel=name, bound="mfma" if mfma_bound else "hbm",
                achieved=round(tfs if mfma_bound else gbs, 2), peak=MFMA_PEAK_TF[dt] if mfma_bound else HBM_PEAK_GBS,
                unit="TFLOP/s" if mfma_bound else "GB/s",
                frac=round((tfs / MFMA_PEAK_TF[dt]) if mfma_bound else (gbs / HBM_PEAK_GBS), 4), traffic=None,
                traffic_source=None, algorithmic_bytes_per_launch=int(d["bytes"] / max(d["launches"], 1)),
                avg_launch_us=round(per_launch_s * 1e6, 2),
                launches_per_step=d["launches"], ms_per_step=round(d["ms"], 4),
                share_of_step=round(d["ms"] / sum(x["ms"] for x in prof.values()), 3))
    # HBM bytes per launch of that kernel family: NOT measured by this run.  It comes from separate rocprofv3 --pmc
    # FETCH_SIZE / WRITE_SIZE passes over this same command (MI355X_MICROARCH.md corrections; tools/summarize_pmc.py);
    # the file names the passes, their time stamps and the git HEAD they were taken at.
    if traffic_file is not None:
        try:
            with open(os.path.join(ROOT, traffic_file)) as fh:
                tj = json.load(fh)
            tr = tj["kernels"].get(name)
            if tr is not None:
                # per launch of THIS line's launch unit (an engine step; a gate-shift site is three kernels)
                roof["traffic"] = (int(tr["hbm_bytes_per_forward"] / max(d["launches"], 1)) if "hbm_bytes_per_forward" in tr
                                   else tr["hbm_bytes_per_launch"])
                roof["traffic_source"] = (f"{traffic_file}: separate rocprofv3 --pmc passes at git {tj.get('git_head')} "
                                          f"({tj.get('fetch_pass', {}).get('mtime')}); not measured by this run")
        except (OSError, KeyError, ValueError):
            pass
    return roof


SGP_FAMILIES = ("sgp_gemm", "sgp_front", "mixer_front", "maxpool", "sgp_fold", "sgp_mlp", "gemm_splitk")


def sgp_roofline(cfg, B, T, eng, plan, sgp_direct, sgp_stage_ms, dt, traffic_file=None):
    """SGP encoder-decoder against the HBM roof with the ALGORITHMIC byte count of SURVEY.md section 8d:
    es * [B*C*Sigma_T + W]: every block/mixer reads its inputs once and writes its output once, weights once."""
    from tdeed_amd.regnet_spec import pyramid_lengths, sgp_up_size
    sgp_steps = [s for s in plan.steps if s.name.startswith("_temp_fine.")]
    n_l, Cc = cfg["n_layers"], eng.pw.spec.feat_dim
    lens = pyramid_lengths(T, n_l)
    sig = sum(2 * lens[i] + lens[i + 1] for i in range(n_l)) + 2 * lens[n_l] \
        + sum((2 * lens[l] + lens[l + 1]) + 2 * lens[l] for l in range(n_l))
    ks_, up_ = cfg["sgp_ks"], sgp_up_size(cfg["sgp_ks"], cfg["sgp_r"])
    Wsgp = (2 * n_l + 1) * (8 * Cc * Cc + (2 * ks_ + up_ + 16) * Cc) + n_l * (14 * Cc * Cc + (4 * ks_ + 2 * up_ + 26) * Cc)
    es_ = 2 if dt == torch.bfloat16 else 4
    sgp_bytes = es_ * (B * Cc * sig + Wsgp)
    sgp_flops = sum(s.flops for s in sgp_steps)
    # stage time: the sub-batches' stage chains timed back to back on one stream (they overlap other work inside the
    # graph; this is the stage's own device time)
    sgp_ms = sum(x[0] for x in sgp_direct)
    # HBM-side traffic of the stage's kernel families per forward, from the committed counter passes (not measured here)
    traffic = None
    if traffic_file is not None:
        try:
            with open(os.path.join(ROOT, traffic_file)) as fh:
                tk = json.load(fh)["kernels"]
            traffic = int(sum(tk[f]["hbm_bytes_per_forward"] for f in SGP_FAMILIES if f in tk and "hbm_bytes_per_forward" in tk[f]))
        except (OSError, KeyError, ValueError):
            traffic = None
    return dict(bound="hbm", algorithmic_bytes=int(sgp_bytes), sigma_T=int(sig), weights=int(Wsgp),
                traffic=traffic, traffic_over_algorithmic=(round(traffic / sgp_bytes, 2) if traffic else None),
                ms=round(sgp_ms, 4), ms_event_sum=round(sgp_stage_ms, 4),
                achieved=round(sgp_bytes / (sgp_ms * 1e-3) / 1e9, 2),
                peak=HBM_PEAK_GBS, unit="GB/s",
                frac=round(sgp_bytes / (sgp_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                launches=len(sgp_steps), launches_per_chain=[x[1] for x in sgp_direct],
                us_per_chain=[round(x[0] * 1e3, 1) for x in sgp_direct],
                chains=("one for the whole batch (behind the sub-batch join)"
                        if getattr(plan, "tail", None) is not None else "one per sub-batch"),
                mfma_tflops=round(sgp_flops / (sgp_ms * 1e-3) / 1e12, 2),
                mfma_frac=round(sgp_flops / (sgp_ms * 1e-3) / 1e12 / MFMA_PEAK_TF[dt], 4))


def kernels_table(prof):
    return {k: dict(ms=round(v["ms"], 4), launches=v["launches"],
                    GBps=round(v["bytes"] / (v["ms"] * 1e-3) / 1e9, 1) if v["ms"] > 0 else 0,
                    TFLOPs=round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 2) if v["ms"] > 0 else 0)
            for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["ms"])}


def infer_sub_record(workload, steps, repeats, depth, rank, dev, traffic_file, streams=None):
    """A driver-visible (d) record of the inference forward of another BASELINE configuration on this GPU (rank 0, N = 1):
    the same execution shape as the headline (one sub-batch, `depth` batches in flight as single-chain HIP graphs on their
    own streams), clips/s, the dominant family's roofline with `traffic` from the committed counter passes, the SGP stage."""
    wl = CONFIGS[workload]
    cfg, B, H, W = wl["cfg"], wl["B"], wl["H"], wl["W"]
    dt = torch.bfloat16
    T = cfg["clip_len"]
    sd = synth.make_state(state_layout.model_state_shapes(cfg), 0)
    # the headline's own streams when the caller has them: new streams created next to the (idle) ones of the headline and the
    # feed measurement can land on a hardware queue one of them already holds (GPU_MAX_HW_QUEUES = 8), and two of the three
    # batches in flight then serialise -- measured 9.16-9.29 vs 8.82 ms per 800MF step
    streams = list(streams[:depth]) if streams is not None and len(streams) >= depth else [torch.cuda.Stream() for _ in range(depth)]
    with torch.cuda.stream(streams[0]):
        eng = ForwardEngine(cfg, sd, dt, dev, use_graph=True, n_split=1)
        plans = [eng.plan(B, H, W, slot=i) for i in range(depth)]
        for i, pl in enumerate(plans):
            eng.set_frames(pl, ops.fill_u8_hash((B, T, 3, H, W), 1000 + rank + 97 * i, dev))
    torch.cuda.synchronize()

    def run(n):
        for i in range(n):
            with torch.cuda.stream(streams[i % depth]):
                eng.run_plan(plans[i % depth])

    run(2 * depth)
    walls, evs = timed_regions(run, steps, repeats, dev, streams)
    el = statistics.median(walls)
    chk = check_timed_outputs(eng, plans, B, T, H, W, rank, dev)
    with torch.cuda.stream(streams[0]):
        prof, sgp_stage_ms = kernel_profile(eng, plans[0])
        sgp_direct = sgp_stage_time(plans[0])
    ms = el / steps * 1e3
    step_bytes, _ = forward_layer_bytes(cfg, B, H, W, dt, dev)
    rec = dict(workload=f"{workload}: {cfg['feature_arch']} + ed_sgp_mixer n_layers={cfg['n_layers']} ks={cfg['sgp_ks']}, "
                        f"L={T}, {H}x{W}, batch {B}/GPU, inference forward, random-init weights",
               value=round(B * steps / el, 2), unit="clips/s", ms_per_step=round(ms, 4), steps=steps, repeats=repeats,
               ms_per_step_repeats=[round(w / steps * 1e3, 4) for w in walls], dtype="bf16", batches_in_flight=depth,
               hip_graph=True, timed_output_check=chk,
               roofline=dominant_roofline(prof, dt, traffic_file),
               roofline_step=dict(bound="hbm", algorithmic_bytes=int(step_bytes),
                                  achieved=round(step_bytes / (ms * 1e-3) / 1e9, 1), peak=HBM_PEAK_GBS, unit="GB/s",
                                  frac=round(step_bytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)),
               roofline_sgp=sgp_roofline(cfg, B, T, eng, plans[0], sgp_direct, sgp_stage_ms, dt, traffic_file),
               kernels=kernels_table(prof))
    del plans, eng
    torch.cuda.empty_cache()
    return rec



# ---------------------------------------------------------------------------------------------- the printed line
# The driver keeps 8 KB of head and 2 KB of tail of a line: the default line is therefore COMPACT (< 7 KB) with the numbers
# a reader looks for hoisted as scalars right behind the contract's fields; `--full` prints every table (what
# tools/prof_r05.sh stores under profiles/), and the full record is also left in gpurun_out/bench_full.json.
_DROP_KEYS = {"note", "kernels", "ms_per_step_repeats", "by_processes", "by_threads", "families_ms", "traffic_source",
              "cpu_model", "measured", "workload", "thread_sweep", "sweep", "repeats", "hip_graph", "steps", "unit_note",
              "bn_bwd_family", "gemm_family", "narrow_bwd_family", "kernel_name", "slots", "algorithmic_bytes_per_launch", "host_cores"}


def _trim(o, drop=_DROP_KEYS, maxstr=110):
    if isinstance(o, dict):
        return {k: _trim(v, drop, maxstr) for k, v in o.items() if k not in drop}
    if isinstance(o, list):
        return [_trim(v, drop, maxstr) for v in o[:8]]
    if isinstance(o, str) and len(o) > maxstr:
        return o[:maxstr - 3] + "..."
    return o


def _g(d, *path, default=None):
    for k in path:
        if not isinstance(d, dict) or k not in d:
            return default
        d = d[k]
    return d


CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config")


def compact_line(out):
    c = {k: out[k] for k in CONTRACT if k in out}
    tr = out.get("train") or {}
    hoist = dict(
        train_rny008_b16_ms_per_step=_g(tr, "rny008_b16", "ms_per_step"),
        train_rny008_b16_clips_per_s=_g(tr, "rny008_b16", "value"),
        train_rny008_b16_step_hbm_GB=(None if _g(tr, "rny008_b16", "roofline_family", "step_hbm_bytes_measured") is None
                                      else round(_g(tr, "rny008_b16", "roofline_family", "step_hbm_bytes_measured") / 1e9, 1)),
        train_rny002_b8_ms_per_step=_g(tr, "rny002_b8", "ms_per_step"),
        infer_800mf_clips_per_s=_g(out, "infer_800mf", "value"),
        infer_800mf_sgp_stage_us=(None if _g(out, "infer_800mf", "roofline_sgp", "ms") is None
                                  else round(_g(out, "infer_800mf", "roofline_sgp", "ms") * 1e3, 1)),
        infer_snb_t250_clips_per_s=_g(out, "infer_snb_t250", "value"),
        sgp_stage_us=(None if _g(out, "roofline_sgp", "ms") is None else round(_g(out, "roofline_sgp", "ms") * 1e3, 1)),
        logit_max_abs_err_fp32=out.get("logit_max_abs_err_fp32"), logit_max_abs_err_bf16=out.get("logit_max_abs_err_bf16"),
        logit_rms_err_bf16=out.get("logit_rms_err_bf16"),
        logit_abs_max=out.get("logit_abs_max"), latency_ms_inflight1=out.get("latency_ms_inflight1"),
        fed_from_host_clips_per_s=_g(out, "fed_from_host", "value"),
        timed_output_check_ok=_g(out, "timed_output_check", "ok"))
    c.update({k: v for k, v in hoist.items() if v is not None})
    c["roofline"] = _trim(out.get("roofline"))
    c["cpu_baseline"] = _trim(out.get("cpu_baseline"))
    if isinstance(c["cpu_baseline"], dict):          # north_star: "core count stated" -- the host's, beside the threads used
        for k in ("host_cores", "cpu_model"):
            if _g(out, "cpu_baseline", k) is not None:
                c["cpu_baseline"][k] = out["cpu_baseline"][k]
    for k in ("roofline_step", "roofline_sgp", "timed_output_check", "fed_from_host", "infer_800mf", "infer_snb_t250", "train",
              "dp_diag"):
        if out.get(k) is not None:
            c[k] = _trim(out[k])
    c["git_head"] = out.get("git_head")
    c["full_record"] = out.get("full_record")
    # never beyond the driver's 8 KB head: shed the optional sub-tables, least important first
    for k in ("fed_from_host", "roofline_step", "timed_output_check"):
        if len(json.dumps(c)) <= 7000:
            break
        c.pop(k, None)
    if len(json.dumps(c)) > 7000:
        c = _trim(c, _DROP_KEYS | {"roofline_step", "config", "sample"}, 60)
    return c


def emit(out, full):
    """rank 0: leave the full record in gpurun_out/ (when writable) and print ONE line"""
    try:
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        path = os.path.join("gpurun_out", "bench_full.json")
        with open(os.path.join(ROOT, path), "w") as f:
            json.dump(out, f)
        out["full_record"] = path
    except OSError:
        out["full_record"] = None
    print(json.dumps(out if full else compact_line(out)))

def _cpu_model():
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return None


def cpu_baseline(engs=None, dev=None):
    """The oracle (our CPU port of the reference forward) timed on this host: BASELINE.json configs[0] = FineDiving_small,
    1 synthetic clip, fp32: a thread sweep {1, 8, 32, 64} -- the reported `value` is the BEST of them
    -- with the CPU model string, the host's core count and the SGP encoder-decoder alone.  The sweep stops at 64 threads:
    with all 256 hardware threads of the GPU box's two EPYC 9575F the oracle's intra-op thread pool oversubscribes and one
    clip took 127 s (round-3 measurement), which would not be a bounded baseline.
    Bounded: 1 pass at 1 thread, warm-up + 2 passes at the other counts (~15 s in all).
    Also the checker of this run's logits: the fp32 and bf16 engines run the same clip with the same weights."""
    from oracle import tdeed_oracle as O
    from tdeed_amd.regnet_spec import regnet_spec
    c = CONFIGS["rny002_b8"]["cfg"]
    sd_np = synth.make_state(state_layout.model_state_shapes(c), 0)
    sd = O.as_torch_state(sd_np)
    clip_np = synth.uint8_clip(1000, (1, 100, 3, 224, 224))
    clip = torch.from_numpy(clip_np)
    spec = regnet_spec(c["feature_arch"])
    ncpu = os.cpu_count() or 1
    have = torch.get_num_threads()
    sweep = {}
    logits = displ = None
    feat = torch.randn(1, 100, spec.feat_dim)
    sgp_sweep = {}
    with torch.no_grad():
        for nt in sorted({1, min(8, ncpu), min(32, ncpu), min(64, ncpu)}):
            torch.set_num_threads(nt)
            ts = []
            for i in range(1 if nt == 1 else 3):
                t0 = time.perf_counter()
                logits, displ, _ = O.forward(clip, sd, c, spec)
                ts.append(time.perf_counter() - t0)
            sweep[nt] = min(ts[1:]) if len(ts) > 1 else ts[0]
            O.ed_sgp_mixer(feat, sd, c["n_layers"], 100)
            t0 = time.perf_counter()
            for _ in range(3):
                O.ed_sgp_mixer(feat, sd, c["n_layers"], 100)
            sgp_sweep[nt] = (time.perf_counter() - t0) / 3
    torch.set_num_threads(have)
    best_nt = min(sweep, key=sweep.get)
    best = sweep[best_nt]
    res = dict(value=round(1.0 / best, 4), unit="clips/s", cores=best_nt, kind="port",
               sample="FineDiving_small forward, 1 synthetic clip (100x3x224x224), fp32; best of a thread sweep (1 pass at 1 "
                      "thread, best of 2 after a warm-up at the other counts)",
               sec_per_clip=round(best, 4), host_cores=ncpu, cpu_model=_cpu_model(),
               thread_sweep_sec_per_clip={str(k): round(v, 4) for k, v in sweep.items()},
               sgp_only_ms_per_clip={str(k): round(v * 1e3, 2) for k, v in sgp_sweep.items()})
    errs = {}
    if dev is not None:
        st = torch.cuda.Stream()
        with torch.cuda.stream(st):
            for name, dt in (("fp32", torch.float32), ("bf16", torch.bfloat16)):
                eng = ForwardEngine(c, sd_np, dt, dev)
                head, _ = eng.forward(clip.to(dev))
                st.synchronize()
                h = head.float().cpu().view(1, 100, -1)
                errs[name] = max(float((h[..., :5] - logits).abs().max()), float((h[..., 5] - displ).abs().max()))
                errs[name + "_rms"] = float((h[..., :5] - logits).pow(2).mean().sqrt())
                del eng
        errs["logit_abs_max"] = float(logits.abs().max())
    return res, errs


def _dist_setup():
    """One process per GPU under torch.distributed.run; RCCL (backend "nccl").  TDEED_DIST_BACKEND=gloo lets several
    ranks share one GPU to exercise the multi-process path on a single-GPU box (functional check, not a measurement)."""
    rank, local, world = tdist.env_world()
    backend = os.environ.get("TDEED_DIST_BACKEND", "nccl")
    ndev = torch.cuda.device_count()
    local_dev = local % max(ndev, 1) if backend != "nccl" else local
    if local_dev >= ndev:
        raise RuntimeError(f"bench.py: rank {rank} of {world} needs GPU {local_dev}, this node shows {ndev} "
                           f"(one process per GPU; TDEED_DIST_BACKEND=gloo lets ranks share a GPU for a functional check)")
    torch.cuda.set_device(local_dev)
    tdist.init(backend=backend, device=torch.device("cuda", local_dev))   # no-op for a single process
    global DIST_INFO
    DIST_INFO = tdist.count_ranks(torch.device("cuda", local_dev))
    if DIST_INFO["ranks"] != world:
        raise RuntimeError(f"the process group counts {DIST_INFO['ranks']} ranks, WORLD_SIZE is {world}")
    return rank, local, world, f"cuda:{local_dev}"


def _dist_fields():
    """config fields that show how many ranks the launch really had: counted by a one-element all-reduce on the process
    group (`rccl_ranks` when that group is RCCL, i.e. backend "nccl"; `dist_ranks` always)."""
    d = dict(dist_backend=DIST_INFO["backend"], dist_ranks=DIST_INFO["ranks"],
             self_launched=os.environ.get("TDEED_BENCH_SELF_LAUNCHED") == "1")
    if DIST_INFO["backend"] == "nccl":
        d["rccl_ranks"] = DIST_INFO["ranks"]
    return d


# ----------------------------------------------------------------------------------------------------- training step
def forward_layer_bytes(cfg, B, H, W, dt, dev):
    """Layer-granular algorithmic bytes of ONE forward of this geometry (every fused layer reads its
    inputs once and writes its output once, weights once): the sum of engine.Step.bytes over an unfused-front plan."""
    from tdeed_amd import engine as E
    sd = synth.make_state(state_layout.model_state_shapes(cfg), 0)
    # layer granularity: the launches that merge several layers of a bottleneck (tdeed_bneck_fwd, tdeed_c1_gconv_fwd) are
    # switched off for this count, so the figure does not move when layers are fused (what a fused launch itself must move is
    # its own Step.bytes, reported per family)
    saved = E.BNECK_ONE_LAUNCH, E.C1_GCONV
    E.BNECK_ONE_LAUNCH = E.C1_GCONV = False
    try:
        eng = ForwardEngine(cfg, sd, dt, dev, use_graph=False, n_split=1)
        plan = eng.plan(B, H, W)
        b, f = sum(s.bytes for s in plan.steps), sum(s.flops for s in plan.steps)
        del eng, plan
    finally:
        E.BNECK_ONE_LAUNCH, E.C1_GCONV = saved
    torch.cuda.empty_cache()
    return b, f


def train_family_roofline(eng, workload, frames, lab, labD, masks, dt):
    """Roofline of the dominant kernel family of the training step, measured live: one EAGER step with every C-ABI call
    bracketed by HIP events on its launch stream (`_lib.profile`), device time summed per entry point.  For the 1x1-conv
    contraction family (tdeed_gemm_fwd: forward + input-gradient GEMMs) the algorithmic bytes come from each call's own
    (M, K, N, residual) arguments: (M K + N K + M N (1 + residual)) * element size.  `traffic` = HBM bytes per launch of the
    same family from the committed counter passes (profiles/r03_train_hbm_traffic.json), null when that file is absent."""
    from tdeed_amd import _lib
    es = 2 if dt == torch.bfloat16 else 4
    for _ in range(2):                                           # keep the device busy: event brackets then hold device time
        eng.step(frames, lab, labD, drop_masks=masks)
    with _lib.profile() as pr:
        eng.step(frames, lab, labD, drop_masks=masks)
    summ = pr.summary()
    total = sum(d["ms"] for d in summ.values())
    name, d = max(summ.items(), key=lambda kv: kv[1]["ms"])
    fam = {k: dict(ms=round(v["ms"], 3), calls=v["calls"], share=round(v["ms"] / total, 3))
           for k, v in sorted(summ.items(), key=lambda kv: -kv[1]["ms"])[:8]}
    rec = dict(kernel=name, family=TRAIN_FAMILY.get(name), measured="one eager step, HIP events around every C-ABI call",
               ms_per_step=round(d["ms"], 3), launches_per_step=d["calls"], share_of_step=round(d["ms"] / total, 3),
               families_ms=fam, bound="hbm", peak=HBM_PEAK_GBS, unit="GB/s", traffic=None, traffic_source=None)
    g = summ.get("tdeed_gemm_fwd")
    if g is not None:
        # tdeed_gemm_fwd(A, lda, A0, lda0, k0, a_scale, a_scale_rows, M, K, N, W, ldw, scale, shift, residual, ...)
        gb = sum((a[7] * a[8] + a[9] * a[8] + a[7] * a[9] * (2 if a[14] else 1)) * es for a in g["args"])
        gf = sum(2.0 * a[7] * a[8] * a[9] for a in g["args"])
        gsec = g["ms"] * 1e-3
        ai = gf / max(gb, 1)
        mfma_bound = ai > MFMA_PEAK_TF[dt] * 1e12 / (HBM_PEAK_GBS * 1e9)
        gemm = dict(ms_per_step=round(g["ms"], 3), launches_per_step=g["calls"], algorithmic_bytes=int(gb), flops=int(gf),
                    GBps=round(gb / gsec / 1e9, 1), TFLOPs=round(gf / gsec / 1e12, 1),
                    frac_hbm=round(gb / gsec / 1e9 / HBM_PEAK_GBS, 4), frac_mfma=round(gf / gsec / 1e12 / MFMA_PEAK_TF[dt], 4),
                    bound="mfma" if mfma_bound else "hbm")
        rec["gemm_family"] = gemm
        if name == "tdeed_gemm_fwd":
            rec.update(bound=gemm["bound"], algorithmic_bytes_per_launch=int(gb / g["calls"]),
                       avg_launch_us=round(g["ms"] / g["calls"] * 1e3, 2),
                       achieved=gemm["TFLOPs"] if mfma_bound else gemm["GBps"],
                       peak=MFMA_PEAK_TF[dt] if mfma_bound else HBM_PEAK_GBS, unit="TFLOP/s" if mfma_bound else "GB/s",
                       frac=gemm["frac_mfma"] if mfma_bound else gemm["frac_hbm"])
    nb = summ.get("tdeed_narrow_conv1_bwd")
    if nb is not None:
        # tdeed_narrow_conv1_bwd(dY [0], Z [1], M [2], Co [3], Ci [4], ..., X [11], Wt [12], R [13], ldr, r_hi [15], r_wi, dX [17],
        # use_mask, bz [19], bmean, bzd [21], ...): conv1's whole backward of a narrow layer in one launch.  Algorithmic bytes =
        # dY and (where it is not recomputed) Z read once, X read once (operand of the weight gradient, of the recomputed z and
        # the sink's mask), the shortcut gradient R (a quarter of the rows behind a stride-2 block), the sink's raw maps bz / bzd,
        # dX written once; the weights and the per-workgroup partials are KBs.
        def nb_bytes(a):
            M, Co, Ci = a[2], a[3], a[4]
            b = M * Co * (1 + (1 if a[1] else 0)) + M * Ci * 2
            if a[13]:
                b += (M // 4 if a[15] else M) * Ci
            b += M * Ci * ((1 if a[19] else 0) + (1 if a[21] else 0))
            return b * es
        nbb = sum(nb_bytes(a) for a in nb["args"])
        nsec = nb["ms"] * 1e-3
        rec["narrow_bwd_family"] = dict(ms_per_step=round(nb["ms"], 3), launches_per_step=nb["calls"], algorithmic_bytes=int(nbb),
                                        GBps=round(nbb / nsec / 1e9, 1), frac_hbm=round(nbb / nsec / 1e9 / HBM_PEAK_GBS, 4))
        if name == "tdeed_narrow_conv1_bwd":
            rec.update(bound="hbm", algorithmic_bytes_per_launch=int(nbb / nb["calls"]),
                       avg_launch_us=round(nb["ms"] / nb["calls"] * 1e3, 2), achieved=round(nbb / nsec / 1e9, 1),
                       peak=HBM_PEAK_GBS, unit="GB/s", frac=round(nbb / nsec / 1e9 / HBM_PEAK_GBS, 4))
    bn = summ.get("tdeed_bn_train_bwd")
    if bn is not None:
        # tdeed_bn_train_bwd(z, dy, y, relu, M, C, ..., dz [13], d_res [14], ...): BatchNorm (batch statistics) backward.
        # Algorithmic bytes = every input map read once (z, dy, and y where the ReLU mask comes from the block output), every
        # output map written once (dz, and d_res where the residual branch takes the masked gradient); the kernel pair reads
        # its inputs twice (statistics pass, then apply pass): that shows in `traffic`, not here.
        bb = sum((2 + (1 if a[2] else 0) + 1 + (1 if a[14] else 0)) * a[4] * a[5] * es for a in bn["args"])
        bsec = bn["ms"] * 1e-3
        rec["bn_bwd_family"] = dict(ms_per_step=round(bn["ms"], 3), launches_per_step=bn["calls"], algorithmic_bytes=int(bb),
                                    GBps=round(bb / bsec / 1e9, 1), frac_hbm=round(bb / bsec / 1e9 / HBM_PEAK_GBS, 4))
        if name == "tdeed_bn_train_bwd":
            rec.update(bound="hbm", algorithmic_bytes_per_launch=int(bb / bn["calls"]),
                       avg_launch_us=round(bn["ms"] / bn["calls"] * 1e3, 2), achieved=round(bb / bsec / 1e9, 1),
                       peak=HBM_PEAK_GBS, unit="GB/s", frac=round(bb / bsec / 1e9 / HBM_PEAK_GBS, 4))
    try:
        with open(os.path.join(ROOT, TRAIN_TRAFFIC_FILE)) as fh:
            tj = json.load(fh)
        if tj.get("workload") == workload:
            tr = tj["kernels"].get(TRAIN_FAMILY.get(name, ""))
            if tr is not None:
                # per launch of the KERNEL family the counters are keyed by (its own launch count, register-stationary and
                # input-gradient forms included) -- not the family's bytes over this C-ABI entry's calls
                rec["traffic"] = int(tr["hbm_bytes_per_launch"])
                rec["traffic_kernel_family"] = dict(name=TRAIN_FAMILY.get(name), launches_per_step=tr.get("kernel_launches_per_step"),
                                                    hbm_bytes_per_step=tr.get("hbm_bytes_per_step"))
                rec["traffic_source"] = (f"{TRAIN_TRAFFIC_FILE}: separate rocprofv3 --pmc passes at git {tj.get('git_head')} "
                                         f"({tj.get('fetch_pass', {}).get('mtime')}); not measured by this run")
            rec["step_hbm_bytes_measured"] = tj.get("step_hbm_bytes")
    except (OSError, KeyError, ValueError):
        pass
    return rec


def train_measure(workload, dtype, steps, warmup, repeats, rank, world, dev, use_graph=True, want_cpu=False,
                  family_roofline=True):
    """Optimisation steps of the training path on synthetic uint8 clips resident in HBM: train-mode forward + CE/MSE loss
    + full backward + (N > 1: gradient all-reduce over RCCL) + fused AdamW; dropout masks fixed."""
    from tdeed_amd.trainer import TrainEngine
    from tdeed_amd.regnet_spec import regnet_spec
    wl = CONFIGS[workload]
    cfg, B, H, W = wl["cfg"], wl["B"], wl["H"], wl["W"]
    dt = torch.bfloat16 if dtype == "bf16" else torch.float32
    T = cfg["clip_len"]
    sd = {k: torch.from_numpy(v) for k, v in synth.make_state(state_layout.model_state_shapes(cfg), 0).items()}
    eng = TrainEngine(cfg, sd, dt, dev, lr=1e-4)
    frames = ops.fill_u8_hash((B, T, 3, H, W), 1000 + rank, dev)
    lab_np, labD_np = synth.labels(5 + rank, B, T, cfg["num_classes"], max(cfg["radi_displacement"], 1))
    lab = torch.from_numpy(lab_np).to(dev)
    labD = torch.from_numpy(labD_np).float().to(dev) if cfg["radi_displacement"] else None
    C = regnet_spec(cfg["feature_arch"]).feat_dim
    masks = [((torch.rand((B, T, C), device=dev) >= 0.5).to(dt) * 2.0) for _ in range(2 if cfg["radi_displacement"] else 1)]
    step = eng.make_step(B, H, W, frames, lab, labD, masks, use_graph=use_graph, world=world)
    for _ in range(max(warmup, 1)):
        step()
    st = torch.cuda.current_stream()
    loss_box = {}

    def run(n):
        for _ in range(n):
            loss_box["l"] = step()
    walls, evs = timed_regions(run, steps, repeats, dev, [st])
    el = statistics.median(walls)
    fb, ff = forward_layer_bytes(cfg, B, H, W, dt, dev)
    ms = el / steps * 1e3
    # algorithmic bytes of a train step at layer granularity: the forward's maps move once in the forward, once in the
    # input-gradient pass (dy in, dx out) and once in the weight-gradient pass (dy and x in): 3x the forward's bytes
    # (optimizer state adds 16 B per parameter).  The same convention as `roofline_step` of the forward.
    tb = 3 * fb + 16 * eng.params.numel
    rec = dict(workload=f"{workload}: {cfg['feature_arch']} + ed_sgp_mixer n_layers={cfg['n_layers']}, L={T}, {H}x{W}, "
                        f"batch {B}/GPU, training step (train-mode fwd + loss + bwd + AdamW), random-init weights",
               value=round(world * B * steps / el, 2), unit="clips/s", ms_per_step=round(ms, 3),
               ms_per_step_repeats=[round(w / steps * 1e3, 3) for w in walls], steps=steps, repeats=repeats, dtype=dtype,
               hip_graph=bool(use_graph), grad_buffer_mb=round(eng.params.numel * 4 / 2 ** 20, 1),
               final_loss=round(float(loss_box["l"][0]), 4),
               parallelism=f"dp{world}" + ((" (%s all-reduce of the flat fp32 gradient buffer, bucketed, overlapped with the "
                                            "backward)" % ("RCCL" if eng.reducer is not None and eng.reducer.backend == "rccl"
                                                           else "torch.distributed")) if world > 1 else ""),
               roofline=dict(bound="hbm", kernel="whole step (layer-granular algorithmic bytes: 3 x forward + optimizer)",
                             algorithmic_bytes=int(tb), achieved=round(tb / (ms * 1e-3) / 1e9, 1), peak=HBM_PEAK_GBS,
                             unit="GB/s", frac=round(tb / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), traffic=None,
                             flops=int(3 * ff), tflops=round(3 * ff / (ms * 1e-3) / 1e12, 1)),
               cpu_baseline=None)
    if rank == 0 and world == 1 and family_roofline:
        try:
            rec["roofline_family"] = train_family_roofline(eng, workload, frames, lab, labD, masks, dt)
        except Exception as e:       # noqa: BLE001  (a side measurement must not take the line down)
            rec["roofline_family"] = dict(error=f"{type(e).__name__}: {e}"[:300])
    if eng.reducer is not None and world > 1:
        # ---- self-diagnosis of the data-parallel step (every rank takes part): the same step with the reduction replaced by
        # a no-op gives the communication the backward does NOT hide; each bucket's collective alone gives what there is to hide
        step_nr = eng.make_step(B, H, W, frames, lab, labD, masks, use_graph=use_graph, world=world, all_reduce=lambda g: None)
        for _ in range(2):
            step_nr()

        def run_nr(n):
            for _ in range(n):
                step_nr()
        walls_nr, _ = timed_regions(run_nr, steps, max(2, repeats - 1), dev, [st])
        ms_nr = statistics.median(walls_nr) / steps * 1e3
        buckets = eng.reducer.measure()
        comm = sum(b_["ms"] for b_ in buckets)
        rec["dp_diag"] = dict(step_ms=round(ms, 3), step_ms_no_reduce=round(ms_nr, 3), exposed_comm_ms=round(ms - ms_nr, 3),
                              comm_alone_ms=round(comm, 3), hidden_frac=(round(1.0 - max(ms - ms_nr, 0.0) / comm, 3) if comm > 0 else None),
                              buckets=buckets,
                              note="exposed = step with the bucketed reduction - the same step with a no-op reduction; "
                                   "comm_alone = each bucket's collective on an idle device (GradReducer.measure)")
    if eng.reducer is not None:                                  # N > 1: which transport reduced the gradients, ranks RCCL saw
        rec["reducer"] = eng.reducer.describe()
        rec["reducer"]["rccl_ranks_expected"] = world
        if "rccl_ranks" in rec["reducer"] and rec["reducer"]["rccl_ranks"] != world:
            raise RuntimeError(f"RCCL communicator has {rec['reducer']['rccl_ranks']} ranks, WORLD_SIZE is {world}")
        if use_graph:
            rec["reducer"]["collectives"] = ("inside the captured step" if eng.last_graph.mode == "one"
                                             else "launched between the two captured halves of the step")
    if want_cpu and rank == 0:
        rec["cpu_baseline"] = cpu_train_baseline()
    del eng, step
    torch.cuda.empty_cache()
    return rec


def decode_rate(B, T, H, W, seconds=3.0):
    """Row f4: JPEG decode throughput of the input pipeline on this host -- feeder.clip_batches with a DecodePool decoding
    whole batches of B clips x T frames (synthetic H x W JPEGs written to a temporary directory, quality 90) straight into
    pinned staging slots.  No GPU work; bounded to ~`seconds`."""
    import tempfile
    import numpy as np
    from PIL import Image
    from tdeed_amd import feeder
    d = tempfile.mkdtemp(prefix="tdeed_frames_")
    rs = np.random.RandomState(0)
    base = rs.randint(0, 256, (H // 8, W // 8, 3), dtype=np.uint8)          # blocky content: a realistic ~10-20 KB per frame
    for i in range(T):
        img = np.kron(np.roll(base, i, axis=1), np.ones((8, 8, 1), dtype=np.uint8))
        img = (img.astype(np.int16) + rs.randint(-12, 13, img.shape)).clip(0, 255).astype(np.uint8)
        Image.fromarray(img).save(os.path.join(d, f"frame{i}.jpg"), quality=90)
    kb = sum(os.path.getsize(os.path.join(d, f)) for f in os.listdir(d)) / T / 1024
    out = {}
    ncpu = os.cpu_count() or 1
    for thr in sorted({min(8, ncpu), min(32, ncpu), min(64, ncpu)}):
        pool = feeder.DecodePool(thr)
        clips = [dict(paths=[d, 0, 0, 0, -1, T], stride=1)] * (B * 64)
        n, t0 = 0, time.perf_counter()
        for batch in feeder.clip_batches(clips, B, (3, H, W), T, pool=pool, depth=2):
            n += B
            if time.perf_counter() - t0 > seconds / 3:
                break
        out[str(thr)] = round(n / (time.perf_counter() - t0), 1)
        pool.close()
    # decode worker PROCESSES writing into shared, page-locked staging slots (feeder.ProcessDecodePool): what scales with the
    # host's cores (JPEG decode in Python threads stops scaling at ~8: the GIL around open / convert / copy)
    outp = {}
    for pr_ in sorted({min(16, ncpu), min(64, ncpu), min(128, ncpu)}):
        try:
            pool = feeder.ProcessDecodePool(pr_)
            clips = [dict(paths=[d, 0, 0, 0, -1, T], stride=1)] * (B * 256)
            n, t0 = 0, None
            for batch in feeder.clip_batches(clips, B, (3, H, W), T, pool=pool, depth=2):
                if t0 is None:                       # the first batch carries the workers' start-up
                    t0 = time.perf_counter()
                    continue
                n += B
                if time.perf_counter() - t0 > seconds / 3:
                    break
            outp[str(pr_)] = round(n / max(time.perf_counter() - t0, 1e-9), 1)
        except Exception as e:       # noqa: BLE001
            outp[str(pr_)] = f"{type(e).__name__}: {e}"[:120]
        finally:
            try:
                pool.close()
            except Exception:        # noqa: BLE001
                pass
    import shutil
    shutil.rmtree(d, ignore_errors=True)
    best = max(out, key=out.get)
    okp = {k: v for k, v in outp.items() if isinstance(v, float)}
    bestp = max(okp, key=okp.get) if okp else None
    return dict(decode_clips_per_s=out[best], threads=int(best), by_threads=out,
                decode_clips_per_s_processes=(okp[bestp] if bestp else None), processes=(int(bestp) if bestp else None),
                by_processes=outp, jpeg_kb_per_frame=round(kb, 1),
                note=f"Pillow (libjpeg) decode of {T} x {H}x{W} JPEG frames per clip into pinned staging slots "
                     "(feeder.clip_batches) by a thread pool (DecodePool) and by worker processes (ProcessDecodePool: "
                     "shared-memory slots page-locked in the consumer); the reference decodes with torchvision.io.read_image "
                     "in 4-8 DataLoader worker processes (train_tdeed.py:131-139)")


def cpu_train_baseline():
    """The oracle's training step (train-mode forward + CE/MSE + autograd backward) on the host cores: the model of BASELINE
    configs[2] (FineDiving_big hyper-parameters: RegNetY-800MF, n_layers 3), ONE synthetic clip of 100 frames at 224x224,
    fp32; bounded to one warm-up-free pass (~10-30 s)."""
    from oracle import tdeed_oracle as O
    from tdeed_amd.regnet_spec import regnet_spec
    have = torch.get_num_threads()
    torch.set_num_threads(min(32, os.cpu_count() or 1))          # the forward sweep's best region (all cores oversubscribe)
    try:
        return _cpu_train_baseline(O, regnet_spec)
    finally:
        torch.set_num_threads(have)


def _cpu_train_baseline(O, regnet_spec):
    c = CONFIGS["rny008_b16"]["cfg"]          # the model of the record this stands beside (FineDiving_big: 800MF, n_layers 3)
    sd0 = O.as_torch_state(synth.make_state(state_layout.model_state_shapes(c), 0))
    par = [k for k in sd0 if state_layout.is_parameter(k)]
    sd = {k: (v.clone().requires_grad_(True) if k in par else v.clone()) for k, v in sd0.items()}
    clip = torch.from_numpy(synth.uint8_clip(1000, (1, 100, 3, 224, 224)))
    lab_np, labD_np = synth.labels(5, 1, 100, c["num_classes"], c["radi_displacement"])
    spec = regnet_spec(c["feature_arch"])
    t0 = time.perf_counter()
    x = O.preprocess(clip, c["crop_dim"])
    f = O.regnet_features(x.reshape(100, *x.shape[2:]), sd, spec, 100, "gsf", training=True).reshape(1, 100, -1)
    enc = O.ed_sgp_mixer(f + sd["temp_enc"][None], sd, c["n_layers"], 100)
    cls, dsp = O.heads(enc, sd, c["radi_displacement"])
    loss = O.loss_fn(cls, torch.from_numpy(lab_np), dsp, torch.from_numpy(labD_np).float())
    loss.backward()
    el = time.perf_counter() - t0
    return dict(value=round(1.0 / el, 4), unit="clips/s", cores=torch.get_num_threads(), kind="port",
                host_cores=os.cpu_count(), cpu_model=_cpu_model(),
                sample="FineDiving_big (RegNetY-800MF + SGP n_layers 3: the model of this record) training step (train-mode "
                       "forward + loss + autograd backward, no optimizer), 1 synthetic clip (100x3x224x224), fp32, one pass",
                sec_per_clip=round(el, 3))


def main_train(a):
    rank, local, world, dev = _dist_setup()
    rec = train_measure(a.workload, a.dtype, a.steps, a.warmup, a.repeats, rank, world, dev, use_graph=not a.no_graph,
                        want_cpu=(world == 1 and not a.no_cpu_baseline))
    if rank == 0:
        wl = CONFIGS[a.workload]
        T, H, W = wl["cfg"]["clip_len"], wl["H"], wl["W"]
        out = dict(metric="clips/sec (L=%d, %dx%d, %s) training step (fwd + loss + bwd + AdamW)" % (T, H, W, a.dtype),
                   value=rec["value"], unit="clips/s", n_gpus=world, steps=a.steps, warmup=a.warmup,
                   ms_per_step=rec["ms_per_step"], higher_is_better=True, scaling="weak", vs_baseline=None,
                   dtype=a.dtype, data="synthetic",
                   config=dict(workload=rec["workload"], clips_per_gpu=wl["B"], parallelism=rec["parallelism"],
                               grad_buffer_mb=rec["grad_buffer_mb"], final_loss=rec["final_loss"], hip_graph=rec["hip_graph"],
                               **_dist_fields()),
                   repeats=a.repeats, ms_per_step_repeats=rec["ms_per_step_repeats"],
                   roofline=rec["roofline"], cpu_baseline=rec["cpu_baseline"], git_head=git_head())
        if "reducer" in rec:
            out["config"]["reducer"] = rec["reducer"]
        if "dp_diag" in rec:
            out["dp_diag"] = rec["dp_diag"]
        print(json.dumps(out))
    if world > 1:
        import torch.distributed as dist
        tdist.barrier()                                          # rank 0's side measurements are over: leave together
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--repeats", type=int, default=5, help="timed regions of --steps steps each; the median is reported")
    ap.add_argument("--workload", default="rny002_b8", choices=list(CONFIGS))
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-train", action="store_true", help="skip the training-step sub-records of the default line")
    ap.add_argument("--no-feed", action="store_true", help="skip the fed-from-pinned-host-memory measurement")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--full", action="store_true",
                    help="print every table (per-kernel tables, sweeps, notes); the default line is compact (< 7 KB: the driver "
                         "keeps 8 KB of a line's head) with the sub-records' headline numbers hoisted as scalars")
    ap.add_argument("--pmc-pass", action="store_true",
                    help="only the timed steps (for rocprofv3 --pmc counter passes: eager launches, no side measurements)")
    ap.add_argument("--split", type=int, default=1,
                    help="sub-batches of whole clips per batch, each on its own stream inside the batch's HIP graph (1: none; "
                         "2 gives the shortest single-batch latency, 1 with three batches in flight the highest throughput)")
    ap.add_argument("--cu-mask", default="none", choices=["none", "interleave", "block"],
                    help="experiment: give each in-flight batch's stream its own share of the CUs (hipExtStreamCreateWithCUMask)")
    ap.add_argument("--inflight", type=int, default=3,
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
    streams = _masked_streams(depth, a.cu_mask) if a.cu_mask != "none" else [torch.cuda.Stream() for _ in range(depth)]
    stream = streams[0]
    with torch.cuda.stream(stream):
        eng = ForwardEngine(cfg, sd, dt, dev, use_graph=not a.no_graph, n_split=a.split)
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
    walls, evs = timed_regions(run, a.steps, a.repeats, dev, streams)
    el = statistics.median(walls)
    if a.pmc_pass:
        if rank == 0:
            print(json.dumps({"pmc_pass": True, "workload": a.workload, "steps": a.steps, "ms_per_step": round(el / a.steps * 1e3, 4),
                              "hip_graph": not a.no_graph, "git_head": git_head()}))
        return
    out_check = check_timed_outputs(eng, plans, B, T, H, W, rank, dev)
    # latency of ONE batch with nothing else in flight (graph replay + sync per step)
    lat = []
    if rank == 0:
        with torch.cuda.stream(stream):
            for _ in range(12):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(stream)
                eng.run_plan(plan)
                e1.record(stream)
                stream.synchronize()
                lat.append(e0.elapsed_time(e1))
    # the same forward fed from pinned host memory: uint8 batches cross PCIe on a copy stream into the idle plan's input
    # buffers while the other plan computes (row f4; `value` above has the clips resident in HBM, this is the rate with the
    # PCIe leg in)
    feed = None
    if world == 1 and depth >= 2 and not a.no_feed:       # single-process side measurement (its timed regions hold barriers)
        hosts = [torch.randint(0, 256, (B, T, 3, H, W), dtype=torch.uint8).pin_memory() for _ in range(3)]
        # H2D lands in a ring of device staging buffers on a copy stream; the forward's first act is a device-to-device copy (1 TB/s, 0.12 ms) staging -> the plan's input
        # buffers, after which the staging slot is free again: the upload of batch i+2 never waits for a forward to END,
        # only for the head of the forward that last used its slot.
        NST = int(os.environ.get("TDEED_FEED_AHEAD", "2")) + 1
        stage = [torch.empty((B, T, 3, H, W), dtype=torch.uint8, device=dev) for _ in range(NST)]
        # copy streams the batch is cut over: ONE.  Alone, two streams carry 56 GB/s against 50; next to three forwards in
        # flight one stream sustains 48 GB/s and two fall to 39 (tools/_feed_ab.sh: 3190 vs 2630 clips/s)
        n_cs = int(os.environ.get("TDEED_FEED_STREAMS", "1"))
        copy_sts = [torch.cuda.Stream() for _ in range(n_cs)]
        ahead = int(os.environ.get("TDEED_FEED_AHEAD", "2"))           # batches uploaded ahead of the forward (<= NST - 1)
        freed = [None] * NST              # event: the D2D copy out of staging slot j has run
        arrived = [None] * NST

        def upload(step):
            j = step % NST
            evs = []
            for k_, cs in enumerate(copy_sts):
                with torch.cuda.stream(cs):
                    if freed[j] is not None:
                        cs.wait_event(freed[j])
                    lo, hi = B * k_ // n_cs, B * (k_ + 1) // n_cs
                    stage[j][lo:hi].copy_(hosts[step % 3][lo:hi], non_blocking=True)
                    ev = torch.cuda.Event()
                    ev.record(cs)
                    evs.append(ev)
            arrived[j] = evs

        def run_fed(n):
            for k0_ in range(min(ahead, n)):
                upload(k0_)
            for i in range(n):
                p_, j = i % depth, i % NST
                pl = plans[p_]
                Bs = B // len(pl.subs)
                with torch.cuda.stream(streams[p_]):
                    for ev in arrived[j]:
                        streams[p_].wait_event(ev)
                    for k_, sb in enumerate(pl.subs):
                        sb.frames.copy_(stage[j][k_ * Bs:(k_ + 1) * Bs].view(Bs * T, 3, H, W), non_blocking=True)
                    ev = torch.cuda.Event()
                    ev.record(streams[p_])
                    freed[j] = ev
                    eng.run_plan(pl)
                if i + ahead < n:
                    upload(i + ahead)

        run_fed(6)
        walls_f, _ = timed_regions(run_fed, a.steps, 3, dev, streams + copy_sts)
        el_f = statistics.median(walls_f)
        feed = dict(value=round(B * a.steps / el_f, 2), unit="clips/s", ms_per_step=round(el_f / a.steps * 1e3, 4),
                    h2d_GBps=round(B * T * 3 * H * W / (el_f / a.steps) / 1e9, 2),
                    note="uint8 clips in pinned host memory -> async H2D on a copy stream into a ring of 3 device staging "
                         "buffers, two batches ahead of the forward that consumes them (its first act: a device-to-device copy "
                         "into the plan's input buffers); 15 MB per clip over PCIe",
                    frac_of_resident=round((B * a.steps / el_f) / (world * B * a.steps / el), 3))
        try:
            feed.update(decode_rate(B, T, H, W))
        except Exception as e:       # noqa: BLE001
            feed["decode_error"] = f"{type(e).__name__}: {e}"[:200]
    with torch.cuda.stream(stream):
        prof, sgp_stage_ms = kernel_profile(eng, plan) if rank == 0 else (None, None)
        sgp_direct = sgp_stage_time(plan) if rank == 0 else None

    if rank == 0:
        ms = el / a.steps * 1e3
        value = world * B * a.steps / el
        roof = dominant_roofline(prof, dt, TRAFFIC_FILE if (a.workload == "rny002_b8" and a.dtype == "bf16") else None)
        sgp_roof = sgp_roofline(cfg, B, T, eng, plan, sgp_direct, sgp_stage_ms, dt,
                                TRAFFIC_FILE if (a.workload == "rny002_b8" and a.dtype == "bf16") else None)
        kernels = kernels_table(prof)
        step_bytes, _ = forward_layer_bytes(cfg, B, H, W, dt, dev)     # layer-granular (fusion-independent), as in round 1 / 2
        step_flops = sum(s.flops for s in plan.steps)
        out = dict(metric=f"clips/sec (L={T}, 224^2, {a.dtype}) forward, per-frame logits", value=round(value, 2),
                   unit="clips/s", n_gpus=world, steps=a.steps, warmup=a.warmup, ms_per_step=round(ms, 4),
                   higher_is_better=True, scaling="weak", vs_baseline=None, dtype=a.dtype, data="synthetic",
                   config=dict(workload=f"{a.workload}: {cfg['feature_arch']} + ed_sgp_mixer n_layers={cfg['n_layers']} "
                                        f"ks={cfg['sgp_ks']}, L={T}, {H}x{W}, batch {B}/GPU, inference forward, "
                                        "random-init weights", clips_per_gpu=B, parallelism=f"dp{world} (clip-sharded, no collective)",
                               hip_graph=not a.no_graph, batches_in_flight=depth,
                               sub_batches_per_batch=len(plan.subs), **_dist_fields()),
                   repeats=a.repeats, ms_per_step_repeats=[round(w / a.steps * 1e3, 4) for w in walls],
                   ms_per_step_hip_events=round(statistics.median(evs) / a.steps, 4),
                   latency_ms_inflight1=round(statistics.median(lat), 4),
                   roofline=roof,
                   roofline_step=dict(bound="hbm", note="whole forward, layer-granular algorithmic bytes (every fused layer "
                                      "reads its inputs and writes its output once) / ms_per_step",
                                      algorithmic_bytes=int(step_bytes), achieved=round(step_bytes / (ms * 1e-3) / 1e9, 1),
                                      peak=HBM_PEAK_GBS, unit="GB/s", frac=round(step_bytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                      tflops=round(step_flops / (ms * 1e-3) / 1e12, 1)),
                   kernels=kernels,
                   roofline_sgp=sgp_roof, timed_output_check=out_check,
                   fed_from_host=feed, cpu_baseline=None, git_head=git_head())
    del plans, plan, eng
    torch.cuda.empty_cache()
    if rank == 0:
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"], errs = cpu_baseline(dev=dev)
            out["logit_max_abs_err_fp32"] = round(errs["fp32"], 7)
            out["logit_max_abs_err_bf16"] = round(errs["bf16"], 5)
            out["logit_abs_max"] = round(errs["logit_abs_max"], 4)
            out["logit_rms_err_fp32"] = round(errs["fp32_rms"], 8)
            out["logit_rms_err_bf16"] = round(errs["bf16_rms"], 5)
    if world == 1 and not a.no_train and a.workload == "rny002_b8" and a.dtype == "bf16":
        # driver-visible forward of the 800MF model (BASELINE configs[2..4] run on it): B = 16 as in configs[2]
        try:
            out["infer_800mf"] = infer_sub_record("rny008_b16", 50, 3, depth, rank, dev, TRAFFIC_FILE_800MF, streams=streams)
        except Exception as e:           # noqa: BLE001  (the headline line must still be printed)
            out["infer_800mf"] = dict(error=f"{type(e).__name__}: {e}"[:300])
        # ... and of the long-clip configuration (BASELINE configs[4] per-GPU share: 800MF, T = 250, B = 4)
        try:
            out["infer_snb_t250"] = infer_sub_record("snb_t250_b4", 40, 3, depth, rank, dev, TRAFFIC_FILE_SNB, streams=streams)
        except Exception as e:           # noqa: BLE001
            out["infer_snb_t250"] = dict(error=f"{type(e).__name__}: {e}"[:300])
        # driver-visible training-step records: BASELINE configs[2] (800MF, B=16) and the 200MF geometry of the headline
        tr = {}
        for wk, st_ in (("rny008_b16", 6), ("rny002_b8", 10)):
            try:
                # the cfg3 record (BASELINE configs[2]) is a full (d) line: cpu_baseline + the dominant family's roofline
                tr[wk] = train_measure(wk, "bf16", st_, 2, 3, rank, world, dev, use_graph=True,
                                       want_cpu=(wk == "rny008_b16" and not a.no_cpu_baseline),
                                       family_roofline=(wk == "rny008_b16"))
            except Exception as e:       # noqa: BLE001  (the headline line must still be printed)
                tr[wk] = dict(error=f"{type(e).__name__}: {e}"[:300])
        out["train"] = tr
    if rank == 0:
        emit(out, a.full)
    if world > 1:
        import torch.distributed as dist
        tdist.barrier()                                          # rank 0's side measurements are over: leave together
        dist.destroy_process_group()


def launch_probe():
    """TDEED_BENCH_LAUNCH_PROBE=1: only the launch path -- rendezvous (gloo, CPU), count the ranks with a collective, rank 0
    prints {"n_gpus": ranks, ...} -- so that `python bench.py --gpus N` (self-launch) and the torchrun form can be tested
    on a box without a GPU.  Not a measurement."""
    rank, local, world = tdist.init(backend="gloo")
    info = tdist.count_ranks()
    tdist.barrier()
    if rank == 0:
        print(json.dumps(dict(launch_probe=True, n_gpus=info["ranks"], world_env=world, dist_backend=info["backend"],
                              requested_gpus=_requested_gpus(sys.argv[1:]),
                              self_launched=os.environ.get("TDEED_BENCH_SELF_LAUNCHED") == "1")))
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    if os.environ.get("TDEED_BENCH_LAUNCH_PROBE") == "1":
        launch_probe()
    else:
        main()
