"""Kernel-layout copies of the master parameters, refreshed by ONE gather launch per dtype.

The training engine keeps packed copies of its weights: bf16 casts, transposes for the input-gradient contractions, MFMA
fragment orders for the grouped / tap convolutions, zero-padded BatchNorm vectors, concatenated depthwise branches.  In
PyTorch terms (what the reference does implicitly under autocast: /root/reference/model/model.py:236-263 casts every weight
per step) each of them is `.to(bf16)` / `.t().contiguous()` / `cat` / an index gather of some master tensor -- ~350 small
launches per step of 3-5 us.  Every one of them is a fixed permutation (with zero holes) of the flat fp32 parameter buffer.

`PackPlan.build(modules)` therefore runs the modules' own `repack()` ONCE in recording mode: the state dict is replaced by
CPU float64 tensors that hold each parameter element's 1-based position in the flat buffer, the helpers below
(`cast_bf16`, `to_bf16`, `transpose`) become value-preserving torch ops that only tag "this copy is bf16" (by adding 2^40),
and whatever tensors `repack()` leaves on the modules are read back as index tables.  Copies that turn out to be plain views
of the master buffer (an ascending run) stay views; everything else lives in one flat bf16 and one flat fp32 buffer at fixed
addresses (stable across HIP-graph replays).  `PackPlan.run()` = `tdeed_gather_cast` x 2, plus a cast and an LDS-tiled
transpose launch for each LARGE dense weight (a transposed copy through an index table reads one 64-byte line per element).
"""
from types import SimpleNamespace

import numpy as np
import torch

from . import ops, ops_bwd as B_
from ._lib import call, ptr, stream_ptr, dtype_code

_BF = float(2 ** 40)
_recording = False


def recording():
    return _recording


# Large dense weights keep their own two launches: a transposed copy through the index table reads one 64-byte line per
# element (measured 6.8 ms for the 140 M packed elements of the 800MF model), an LDS-tiled transpose streams.  Below this
# size the gather is cheaper than a launch.
DIRECT_MIN = 65536
import os as _os
# the large weights' casts / transposes as one table-driven launch (MULTI_DIRECT = False: one cast + one transpose launch each)
MULTI_DIRECT = True
_plan = None                                                     # the PackPlan being recorded


def _run_of(w):
    """(flat offset, n) if the index tensor w is an untagged ascending run (a plain view of the master buffer), else None."""
    v = w.reshape(-1)
    n = v.numel()
    first = float(v[0])
    if first < 1 or first >= _BF or float(v[-1]) != first + n - 1:
        return None
    if not torch.equal(v, torch.arange(n, dtype=torch.float64) + first):
        return None
    return int(first) - 1, n


def cast_bf16(w):
    """fp32 -> bf16 copy (recording: tag the positions; large tensors: a recorded cast launch into a device tensor)."""
    if _recording:
        if w.dtype == torch.float64 and w.numel() >= DIRECT_MIN and _plan is not None and _run_of(w) is not None:
            off, n = _run_of(w)
            out = torch.empty(tuple(w.shape), dtype=torch.bfloat16, device=_plan.device)
            _plan.direct.append(("cast", off, n, out))
            _plan.direct_out.add(id(out))
            return out
        return torch.where(w != 0, w + _BF, w)
    return ops.cast_bf16(w)


def to_bf16(t):
    if _recording:
        return torch.where(t != 0, torch.where(t >= _BF, t, t + _BF), t)
    return t.to(torch.bfloat16)


def transpose(w):
    """[N][K] -> [K][N] copy."""
    if _recording:
        if _plan is not None and id(w) in _plan.direct_out:      # a recorded large cast: transpose it with its own launch
            out = torch.empty((w.shape[1], w.shape[0]), dtype=w.dtype, device=w.device)
            _plan.direct.append(("transpose", w, out))
            _plan.direct_out.add(id(out))
            return out
        if w.numel() >= DIRECT_MIN and _plan is not None and _run_of(w) is not None:      # fp32 mode: from the master view
            off, n = _run_of(w)
            src = _plan.params.flat[off:off + n].view(tuple(w.shape))
            out = torch.empty((w.shape[1], w.shape[0]), dtype=torch.float32, device=_plan.device)
            _plan.direct.append(("transpose", src, out))
            _plan.direct_out.add(id(out))
            return out
        return w.t().contiguous()
    return B_.transpose(w)


def pack_ws(w):
    """[N][K] dense weight (N, K multiples of 32) -> the fragment order of the weight-/register-stationary contractions
    ([N/16][K/32][64][8], engine.pack_ws_weights): a pure permutation, so it records like any other copy (apply it to the
    fp32 master view and cast afterwards: `to_bf16(pack_ws(w))`)."""
    N, K = w.shape
    if N % 32 or K % 32:
        raise ValueError(f"pack_ws: N={N}, K={K} must be multiples of 32")
    NT, KS = N // 16, K // 32
    n = torch.arange(16)
    rows = torch.cat([32 * (nt // 2) + 8 * (n // 4) + 4 * (nt % 2) + (n % 4) for nt in range(NT)]).to(w.device)
    fr = w[rows].reshape(NT, 16, KS, 4, 8).permute(0, 2, 3, 1, 4).contiguous()
    return fr.reshape(NT, KS, 64, 8)


def pad1d(v, n):
    """v (F,) -> (n,) with zeros behind."""
    if v.numel() == n:
        return v.reshape(n)
    return torch.cat([v.reshape(-1), torch.zeros(n - v.numel(), dtype=v.dtype, device=v.device)])


class _IndexState(dict):
    """state dict stand-in while recording: parameters are their own flat positions; anything else must not be packed."""

    def __missing__(self, k):
        raise KeyError(f"repack() read '{k}', which is not a parameter of the flat buffer")


class PackPlan:
    def __init__(self, params, device):
        self.params, self.device = params, torch.device(device)
        self.bufs = {}
        self.tables = {}
        self.direct = []                                          # large dense weights: ("cast", off, n, out) / ("transpose", src, out)
        self.direct_out = set()                                   # ids of their output tensors (final as they are)
        self.n_alias = self.n_packed = 0

    # ------------------------------------------------------------------ recording
    def build(self, modules):
        """modules: objects with `.sd` (the shared state dict) and `.repack()`; nested objects that hold their own `.sd`
        (GateShiftTrain) are switched with their owner."""
        global _recording, _plan
        flat = self.params
        idx_sd = _IndexState()
        for k, (o, n) in flat.index.items():
            idx_sd[k] = (torch.arange(n, dtype=torch.float64) + float(o + 1)).view(tuple(flat.shapes[k]))
        owners = []

        def holders(obj, seen):
            if id(obj) in seen or not hasattr(obj, "__dict__"):
                return
            seen.add(id(obj))
            if isinstance(getattr(obj, "sd", None), dict):
                owners.append(obj)
            for v in vars(obj).values():
                if hasattr(v, "__dict__") and not isinstance(v, (torch.Tensor, SimpleNamespace)):
                    holders(v, seen)

        seen = set()
        for m in modules:
            holders(m, seen)
        saved = [(o, o.sd) for o in owners]
        _recording, _plan = True, self
        try:
            for o in owners:
                o.sd = idx_sd
            for m in modules:
                m.repack()
        finally:
            _recording, _plan = False, None
            for o, sd in saved:
                o.sd = sd
        # ---- read the index tensors back
        found = []                                                # (setter, tensor)
        done = {}

        def walk(obj, depth=0):
            if depth > 6:
                return
            if isinstance(obj, list):
                items = [(i, v, (lambda o, i: (lambda x: o.__setitem__(i, x)))(obj, i)) for i, v in enumerate(obj)]
            elif hasattr(obj, "__dict__") and not isinstance(obj, torch.Tensor):
                items = [(k, v, (lambda o, k: (lambda x: setattr(o, k, x)))(obj, k)) for k, v in vars(obj).items()
                         if k not in ("sd", "ctx", "blk", "state", "params")]
            else:
                return
            for _, v, setter in items:
                if isinstance(v, torch.Tensor):
                    if v.device.type == "cpu" and id(v) not in self.direct_out:
                        found.append((setter, v))
                elif isinstance(v, (list, SimpleNamespace)) or (hasattr(v, "__dict__") and not callable(v)):
                    if id(v) not in done:
                        done[id(v)] = True
                        walk(v, depth + 1)

        for m in modules:
            walk(m)
        segs = {torch.bfloat16: [], torch.float32: []}
        sizes = {torch.bfloat16: 0, torch.float32: 0}
        todo = []
        cache = {}
        for setter, t in found:
            if id(t) in cache:
                todo.append((setter,) + cache[id(t)])
                continue
            if t.dtype != torch.float64:                           # a constant made on the recording device
                ent = ("const", t.to(self.device), None)
            else:
                v = t.reshape(-1).numpy()
                tagged = v >= _BF
                nz = v != 0
                if tagged.any() and not (tagged == nz).all():
                    raise RuntimeError("repack: a packed tensor mixes bf16 and fp32 sources")
                pos = np.where(tagged, v - _BF, v).astype(np.int64)
                n = pos.size
                if (not tagged.any()) and n > 0 and pos[0] >= 1 and np.array_equal(pos, pos[0] + np.arange(n)):
                    ent = ("alias", int(pos[0] - 1), tuple(t.shape))
                    self.n_alias += 1
                else:
                    dt = torch.bfloat16 if tagged.any() else torch.float32
                    off = sizes[dt]
                    npad = (n + 7) // 8 * 8
                    tab = np.zeros(npad, np.int32)
                    tab[:n] = pos
                    segs[dt].append(tab)
                    sizes[dt] += npad
                    ent = ("packed", (dt, off, n), tuple(t.shape))
                    self.n_packed += 1
            cache[id(t)] = ent
            todo.append((setter,) + ent)
        for dt in segs:
            if sizes[dt]:
                self.tables[dt] = torch.from_numpy(np.concatenate(segs[dt])).to(self.device)
                self.bufs[dt] = torch.zeros(sizes[dt], dtype=dt, device=self.device)
        for setter, kind, a, shape in todo:
            if kind == "const":
                setter(a)
            elif kind == "alias":
                n = int(np.prod(shape)) if shape else 1
                setter(flat.flat[a:a + n].view(shape))
            else:
                dt, off, n = a
                setter(self.bufs[dt][off:off + n].view(shape))
        self.run()
        return self

    # ------------------------------------------------------------------ steady state
    def _direct_table(self):
        """The recorded large-weight casts / transposes as ONE table for tdeed_multi_cast_transpose: every op must be a cast of a
        master run (bf16 mode) optionally followed by transposes of its output, or a transpose of a master view (fp32 mode);
        anything else keeps its own launch.  Returns (device table, entries, tiles, dtype code, leftover ops)."""
        ents, left = {}, []
        flat = self.params.flat
        for op in self.direct:
            if op[0] == "cast":
                _, off, n, out = op
                C = out.shape[-1]
                ents[id(out)] = dict(src=flat.data_ptr() + 4 * off, R=n // C, C=C, dst=out.data_ptr(), dstT=0, dt=out.dtype)
        for op in self.direct:
            if op[0] != "transpose":
                continue
            _, src, out = op
            e = ents.get(id(src))
            if e is not None and e["dstT"] == 0 and src.dim() == 2:
                e["dstT"] = out.data_ptr()
            elif (src.dtype == torch.float32 and src.dim() == 2 and src.is_contiguous()
                  and flat.data_ptr() <= src.data_ptr() < flat.data_ptr() + 4 * flat.numel()):
                ents[id(out)] = dict(src=src.data_ptr(), R=src.shape[0], C=src.shape[1], dst=0, dstT=out.data_ptr(), dt=out.dtype)
            else:
                left.append(op)
        rows, tiles = [], 0
        dts = {e["dt"] for e in ents.values()}
        if len(dts) != 1:
            return None, 0, 0, 0, list(self.direct)
        for e in ents.values():
            rows.append((e["src"], e["R"], e["C"], e["dst"], e["dstT"], tiles))
            tiles += ((e["R"] + 31) // 32) * ((e["C"] + 31) // 32)
        tab = torch.tensor(rows, dtype=torch.int64).to(self.device)
        return tab, len(rows), tiles, dtype_code(dts.pop()), left

    def run(self):
        for dt, tab in self.tables.items():
            call("tdeed_gather_cast", ptr(self.params.flat), ptr(tab), tab.numel(), ptr(self.bufs[dt]), dtype_code(dt),
                 stream_ptr())
        if self.direct and MULTI_DIRECT:
            if getattr(self, "_dt", None) is None:
                self._dt = self._direct_table()
            tab, n, tiles, dc, left = self._dt
            if n:
                call("tdeed_multi_cast_transpose", ptr(tab), n, tiles, dc, stream_ptr())
            ops_ = left
        else:
            ops_ = self.direct
        for op in ops_:
            if op[0] == "cast":
                _, off, n, out = op
                call("tdeed_cast_f32_to_bf16", ptr(self.params.flat[off:off + n]), ptr(out), n, stream_ptr())
            else:
                _, src, out = op
                call("tdeed_transpose", ptr(src), src.shape[0], src.shape[1], ptr(out), dtype_code(src.dtype), stream_ptr())
