"""PARKED (round 4, measured slower): one temporal stage per GROUP of batches in flight.

VERDICT r3 item 4 asked for the SGP encoder-decoder + heads to run once per group of in-flight batches (24 clips instead of
3 x 8): built as three methods of ForwardEngine (below, verbatim; bind them with `attach(ForwardEngine)`), tested (every
slot's logits equal the B=1 forwards, partial groups flushed) and measured on MI355X, cfg2, bf16, same box, bench.py with
`--inflight D`:

    shape                                   D=2      D=3      D=4     clips/s
    per-batch temporal stage (product)     4366     4675      --
    one stage per group (this file)        4525     4064     4327

The stage itself does get cheaper (one chain over 24 clips: 308 us = 103 us per 8-clip batch against 202 us), but three
TRUNK graphs running against each other all the time are slower than three whole forwards: trunks alone, no temporal
stage at all, take 1.88 ms per step against 1.72 ms for complete forwards in the per-batch shape -- the latency-bound
stage of one batch is what lets the other two batches' bandwidth-bound launches run with less contention.  A first
version that staged the features through one shared buffer (a join of all three streams per group) gave 4163.
"""
import os
from types import SimpleNamespace

import torch

from tdeed_amd import _lib
from tdeed_amd.engine import _drain_dead_graphs
from tdeed_amd.streams import new_stream


def plan_group(self, B, H, W, depth=3, flip=False):
    """`depth` batches of B clips in flight (each its own buffer set and single-chain HIP graph, replayed on the caller's
    stream of that slot) whose temporal stage -- SGP encoder-decoder + heads, a chain of ~16 latency-bound launches whose
    cost is dominated by its 19 MB of weights and its launch boundaries, not by the row count -- runs ONCE per group over
    all depth * B clips on a stream of its own (model/modules.py:69-87 is independent per clip).  Slot i's trunk leaves its
    pooled features in a buffer of its own; behind the trunk its stream copies them (0.6 MB) into rows [i*B, (i+1)*B) of
    the group's feature buffer.  There are TWO group buffers (and two tail graphs) used alternately, so the trunks of
    group g + 1 never wait for group g's stage: the streams are never joined, the stage of one group overlaps the trunks
    of the next.  Every batch still gets the complete forward; after run_group_slot(.., depth - 1) or flush_group the
    group's logits are `grp.head_out` (rows [i*B*T, (i+1)*B*T) for slot i) once `grp.tail_done` has passed."""
    key = ("group", B, H, W, bool(flip), depth)
    if key in self._plans:
        return self._plans[key]
    pw = self.pw
    T, C, dev = pw.clip_len, pw.spec.feat_dim, self.device
    N = depth * B
    f_slot = [torch.empty((B, T, C), dtype=self.act_dtype, device=dev) for _ in range(depth)]
    r_slot = [torch.empty((B * T, 2), dtype=torch.float32, device=dev) for _ in range(depth)]
    subs = [self._build(B, H, W, bool(flip), set(), feat_out=f_slot[i], feat_rs=r_slot[i]) for i in range(depth)]
    feats = [torch.empty((N, T, C), dtype=self.act_dtype, device=dev) for _ in range(2)]
    frss = [torch.empty((N * T, 2), dtype=torch.float32, device=dev) for _ in range(2)]
    heads = [torch.empty((N * T, pw.n_out), dtype=torch.float32, device=dev) for _ in range(2)]
    tails = []
    for par in range(2):
        feats[par]._td_rowstat = frss[par]
        tails.append(self._build_tail(N, feats[par], heads[par]))
    grp = SimpleNamespace(subs=subs, tails=tails, tail=tails[0], depth=depth, B=B, T=T, f_slot=f_slot, r_slot=r_slot,
                          feats=feats, frss=frss, heads=heads, head_out=heads[0], par=0, graphs=None, tail_graphs=None,
                          tail_stream=new_stream(dev), trunk_ev=[[torch.cuda.Event() for _ in range(depth)] for _ in range(2)],
                          tail_done_par=[None, None], tail_done=None, pending=0, graph=None,
                          steps=[st for sb in subs for st in sb.steps] + tails[0].steps,
                          pool_bytes=sum(sb.pool_bytes for sb in subs) + sum(t_.pool_bytes for t_ in tails))
    self._plans[key] = grp
    return grp

def set_group_frames(self, grp, i, frames_u8):
    """Copy a (B,T,3,H,W) uint8 batch into slot i's input buffer."""
    grp.subs[i].frames.copy_(frames_u8.reshape(-1, *frames_u8.shape[2:]), non_blocking=True)

def run_group_slot(self, grp, i):
    """Issue slot i's trunk on the CURRENT stream (HIP-graph replay; eager launches with use_graph=False) and the copy
    of its features into the current group buffer; behind the group's last slot, the temporal stage on the group's tail
    stream (flush_group)."""
    st = torch.cuda.current_stream()
    if self.use_graph and st.cuda_stream == 0:
        raise RuntimeError("graph replay needs a non-default stream: wrap the call in torch.cuda.stream(s)")
    if self.use_graph and grp.graphs is None:
        _drain_dead_graphs()
        for sb in grp.subs:                                   # warm-up launches (module load, argument validation)
            for s_ in sb.steps:
                s_.fn()
        for tl in grp.tails:
            for s_ in tl.steps:
                s_.fn()
        st.synchronize()
        grp.graphs = [self._capture(st, lambda sb=sb: [x.fn() for x in sb.steps]) for sb in grp.subs]
        grp.tail_graphs = [self._capture(st, lambda tl=tl: [x.fn() for x in tl.steps]) for tl in grp.tails]
        grp.graph = SimpleNamespace(subs=list(grp.graphs) + [grp.tail_graphs[1]], tail=grp.tail_graphs[0])   # (for __del__)
        st.synchronize()
    if self.use_graph:
        _lib.call("tdeed_graph_launch", grp.graphs[i], st.cuda_stream)
    else:
        for s_ in grp.subs[i].steps:
            s_.fn()
    par, B, T = grp.par, grp.B, grp.T
    if grp.tail_done_par[par] is not None:
        st.wait_event(grp.tail_done_par[par])                 # the stage that last read this group buffer (two groups ago)
    if os.environ.get("TDEED_GROUP_NOCOPY") != "1":
        grp.feats[par][i * B:(i + 1) * B].copy_(grp.f_slot[i], non_blocking=True)
        grp.frss[par][i * B * T:(i + 1) * B * T].copy_(grp.r_slot[i], non_blocking=True)
    grp.trunk_ev[par][i].record(st)
    grp.pending = max(grp.pending, i + 1)
    if i == grp.depth - 1:
        self.flush_group(grp)

def flush_group(self, grp):
    """Run the temporal stage over the slots issued since the last one (all `depth * B` rows are computed; rows of slots
    that were not issued this time hold whatever that group buffer held).  Called by run_group_slot behind the last
    slot, and by the caller when it stops in the middle of a group."""
    if grp.pending == 0:
        return
    ts, par = grp.tail_stream, grp.par
    for ev in grp.trunk_ev[par][:grp.pending]:
        ts.wait_event(ev)
    with torch.cuda.stream(ts):
        if os.environ.get("TDEED_GROUP_NOTAIL") == "1":
            pass
        elif self.use_graph:
            _lib.call("tdeed_graph_launch", grp.tail_graphs[par], ts.cuda_stream)
        else:
            for s_ in grp.tails[par].steps:
                s_.fn()
        ev = torch.cuda.Event()
        ev.record(ts)
    grp.tail_done_par[par] = grp.tail_done = ev
    grp.head_out = grp.heads[par]
    grp.pending = 0
    grp.par ^= 1



def attach(cls):
    """Bind the three methods to ForwardEngine (experiments only)."""
    cls.plan_group, cls.set_group_frames, cls.run_group_slot, cls.flush_group = (plan_group, set_group_frames, run_group_slot,
                                                                                 flush_group)
