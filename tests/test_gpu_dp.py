"""Data-parallel training on the GPU box: the RCCL communicator behind the C ABI (single rank), and a 2-process job that
shares the one GPU over gloo (TDEED_DIST_BACKEND=gloo style): the DP step must equal AdamW on the MEAN of the per-rank
single-process gradients (SURVEY.md section 8e "DP parity test").  -m gpu only."""
import ctypes
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from helpers import t
from tdeed_amd import synth, state_layout

pytestmark = pytest.mark.gpu
DEV = "cuda"
CFG = dict(feature_arch="rny002_gsf", clip_len=6, crop_dim=None, n_layers=2, sgp_ks=5, sgp_r=2, num_classes=3,
           radi_displacement=2)


def test_rccl_communicator_single_rank_all_reduce_and_join():
    """tdeed_comm_* through ctypes: unique id, init (world 1), in-place sum on the communicator's stream, join, also inside a
    captured HIP graph (the form the training step uses)."""
    from tdeed_amd import _lib
    raw = (ctypes.c_ubyte * 128)()
    _lib.call("tdeed_comm_unique_id", raw)
    h = ctypes.c_void_p()
    _lib.call("tdeed_comm_init", ctypes.byref(h), raw, 1, 0)
    w, r = ctypes.c_int(), ctypes.c_int()
    _lib.call("tdeed_comm_info", h, ctypes.byref(w), ctypes.byref(r))
    assert (w.value, r.value) == (1, 0)
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        x = torch.arange(1 << 20, dtype=torch.float32, device=DEV)
        ref = x.clone()
        x.mul_(2.0)
        _lib.call("tdeed_comm_all_reduce", h, x.data_ptr(), x.numel(), 0, st.cuda_stream)
        _lib.call("tdeed_comm_all_reduce_rs_ag", h, x.data_ptr(), x.numel(), 0, st.cuda_stream)
        _lib.call("tdeed_comm_join", h, st.cuda_stream)
        y = x + 1.0
        st.synchronize()
    assert torch.equal(y, ref * 2.0 + 1.0)
    xb = torch.ones(4096, dtype=torch.bfloat16, device=DEV)
    with torch.cuda.stream(st):
        _lib.call("tdeed_comm_all_reduce", h, xb.data_ptr(), xb.numel(), 1, st.cuda_stream)
        _lib.call("tdeed_comm_join", h, st.cuda_stream)
        st.synchronize()
    assert float(xb.float().sum()) == 4096.0
    _lib.call("tdeed_comm_destroy", h)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _batch(rank):
    B, T = 2, CFG["clip_len"]
    frames = t(synth.uint8_clip(1300 + rank, (B, T, 3, 64, 64)))
    lab, labD = synth.labels(1400 + rank, B, T, CFG["num_classes"], CFG["radi_displacement"], fg_frac=0.4)
    return frames, t(lab).long(), t(labD).float()


def _dp_worker(rank, world, port, q, graph=False):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch.distributed as dist
    from tdeed_amd import dist as tdist
    from tdeed_amd.trainer import TrainEngine
    torch.cuda.set_device(0)                                  # both ranks share the one GPU of the box
    tdist.init(backend="gloo")
    sd = {k: t(v) for k, v in synth.make_state(state_layout.model_state_shapes(CFG), 17).items()}
    eng = TrainEngine(CFG, sd, act_dtype=torch.float32, device="cuda:0", lr=1e-3)
    red = eng.set_reducer("auto")
    assert red is not None and red.backend == "torch" and red.world == world and len(red.buckets) == 2
    frames, lab, labD = _batch(rank)
    if graph:
        # the captured form: gloo cannot be captured, so the step is two graphs with bucket 0's all-reduce launched between
        # them (the same structure a capture failure of the RCCL collectives falls back to)
        step = eng.make_step(frames.shape[0], 64, 64, frames.to("cuda:0"), lab.to("cuda:0"), labD.to("cuda:0"), None,
                             use_graph=True, world=world)
        assert eng.last_graph.mode == "two" and eng.last_graph.graph_b is not None
        step()
    else:
        eng.step(frames.to("cuda:0"), lab.to("cuda:0"), labD.to("cuda:0"))
    torch.cuda.synchronize()
    keys = [k for k in eng.params.index]
    q.put((rank, {k: eng.state[k].detach().cpu().numpy() for k in keys},
           eng.params.grad.detach().cpu().numpy()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("graph", [False, True])
def test_two_rank_dp_step_equals_adamw_on_the_mean_of_per_rank_gradients(graph):
    from tdeed_amd.trainer import TrainEngine
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_dp_worker, args=(r, 2, port, q, graph)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=300) for _ in procs), key=lambda x: x[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    # single-process reference: per-rank gradients from fresh engines, averaged, one AdamW step
    grads = []
    for rank in range(2):
        sd = {k: t(v) for k, v in synth.make_state(state_layout.model_state_shapes(CFG), 17).items()}
        eng = TrainEngine(CFG, sd, act_dtype=torch.float32, device=DEV, lr=1e-3)
        frames, lab, labD = _batch(rank)
        eng.accumulate(frames.to(DEV), lab.to(DEV), labD.to(DEV))
        grads.append(eng.params.grad.clone())
    mean = (grads[0] + grads[1]) * 0.5
    # both ranks hold the summed gradient buffer and identical parameters afterwards
    assert np.array_equal(res[0][2], res[1][2])
    assert float((torch.from_numpy(res[0][2]).to(DEV) * 0.5 - mean).abs().max()) <= 1e-6 * float(mean.abs().max()) + 1e-12
    eng.params.grad.copy_(mean)
    eng.apply()
    torch.cuda.synchronize()
    for k in eng.params.index:
        want = eng.state[k].detach().cpu().numpy()
        for r in (0, 1):
            assert np.abs(res[r][1][k] - want).max() <= 2e-7 + 1e-6 * np.abs(want).max(), (k, r)



@pytest.mark.parametrize("launcher", ["self", "torchrun"])
@pytest.mark.parametrize("mode", ["infer", "train"])
def test_bench_runs_as_two_ranks_on_one_gpu(mode, launcher):
    """Both launch forms of a 2-rank job, two ranks sharing this box's GPU over gloo: "self" = plain `python bench.py --gpus 2`
    (bench.py starts its own child ranks before anything touches the GPU), "torchrun" = the driver's
    `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...`.  Every rank must reach every collective
    (barriers, max over ranks, gradient buckets), rank 0 must print ONE JSON line for the whole job, and the line must show
    the rank count the process group itself counted."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, TDEED_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    tail = [os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--repeats", "2", "--no-cpu-baseline"]
    if launcher == "self":
        cmd = [sys.executable] + tail
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
               "--master-port", str(_free_port())] + tail
    if mode == "train":
        cmd += ["--mode", "train", "--workload", "rny002_b8"]
    r = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["scaling"] == "weak" and rec["value"] > 0
    assert rec["config"]["clips_per_gpu"] == 8
    assert rec["config"]["dist_ranks"] == 2 and rec["config"]["dist_backend"] == "gloo"
    assert rec["config"]["self_launched"] == (launcher == "self")
    if mode == "train":
        red = rec["config"]["reducer"]
        assert red["world"] == 2 and len(red["buckets_mb"]) == 2 and "between the two captured halves" in red["collectives"]
        dg = rec["dp_diag"]          # the self-diagnosis of an N > 1 training line
        assert len(dg["buckets"]) == 2 and all(b["ms"] > 0 for b in dg["buckets"]) and dg["step_ms_no_reduce"] > 0
        assert abs(dg["exposed_comm_ms"] - (dg["step_ms"] - dg["step_ms_no_reduce"])) < 1e-2
