"""Deterministic synthetic weights / clips / labels.

A counter-based generator (splitmix64 of ``seed ^ fnv1a(name) + index``) written
with integer arithmetic only, so the golden-vector script (which fills the
*reference* model in the build container) and the tests / bench on the GPU box
produce bit-identical tensors without sharing any RNG state and without
depending on libm.  There are no checkpoints or datasets on either machine
(SURVEY.md section 2 rows 14-15), so every parity case starts from here.
"""
import re
import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def fnv1a64(name: str) -> int:
    h = 0xCBF29CE484222325
    for b in name.encode("utf-8"):
        h ^= b
        h = (h * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return h


def _splitmix64(x: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
        z = x
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
        return z ^ (z >> np.uint64(31))


def hash_u64(seed: int, name: str, n: int, offset: int = 0) -> np.ndarray:
    base = np.uint64((seed * 0x9E3779B97F4A7C15 ^ fnv1a64(name)) & 0xFFFFFFFFFFFFFFFF)
    idx = np.arange(offset, offset + n, dtype=np.uint64)
    with np.errstate(over="ignore"):
        return _splitmix64(_splitmix64(idx ^ base) + idx)


def uniform01(seed: int, name: str, n: int) -> np.ndarray:
    """float64 in [0,1) with 32 bits of resolution (exact)."""
    h = hash_u64(seed, name, n)
    return (h >> np.uint64(32)).astype(np.float64) * (1.0 / 4294967296.0)


def normalish(seed: int, name: str, n: int) -> np.ndarray:
    """Zero-mean unit-variance bell (Irwin-Hall of four 16-bit uniforms), exact arithmetic."""
    h = hash_u64(seed, name, n)
    s = np.zeros(n, dtype=np.int64)
    for k in range(4):
        s += ((h >> np.uint64(16 * k)) & np.uint64(0xFFFF)).astype(np.int64)
    # each u ~ U{0..65535}: mean 32767.5, var (65536^2-1)/12
    return (s.astype(np.float64) - 4 * 32767.5) * (1.0 / np.sqrt(4 * (65536.0 ** 2 - 1) / 12.0))


def uint8_clip(seed: int, shape) -> np.ndarray:
    """Uniform {0..255} clip, e.g. shape (B, T, 3, H, W)."""
    n = int(np.prod(shape))
    out = np.empty(n, dtype=np.uint8)
    step = 1 << 22
    for lo in range(0, n, step * 8):
        m = min(step * 8, n - lo)
        nw = (m + 7) // 8
        h = hash_u64(seed, "clip", nw, offset=lo // 8)
        out[lo:lo + m] = h.view(np.uint8)[:m]
    return out.reshape(shape)


def labels(seed: int, B: int, T: int, num_classes: int, radi: int, fg_frac: float = 0.05):
    """int64 labels (0 = background, ~5 % foreground) and integer displacements in [-radi, radi]."""
    u = uniform01(seed, "label_fg", B * T)
    c = (hash_u64(seed, "label_cls", B * T) % np.uint64(max(num_classes, 1))).astype(np.int64) + 1
    lab = np.where(u < fg_frac, c, 0).reshape(B, T)
    d = (hash_u64(seed, "label_d", B * T) % np.uint64(2 * radi + 1)).astype(np.int64) - radi
    return lab, d.reshape(B, T)


# --------------------------------------------------------------------------- weights
_DW_KEYS = ("psi", "fc", "convw", "convkw", "global_fc")


def _kind(key: str, shape) -> tuple:
    """(distribution, a, b) for a state_dict entry of the reference's key grammar."""
    last = key.rsplit(".", 1)[-1]
    if last == "num_batches_tracked":
        return ("zero", 0, 0)
    if last == "running_mean":
        return ("normal", 0.0, 0.1)
    if last == "running_var":
        return ("uniform", 0.5, 1.5)
    if key == "temp_enc":
        return ("normal", 0.0, 1.0 / shape[0])
    is_norm = re.search(r"(\.bn\.|\.ln\d?\.|\.gn\.)", "." + key) is not None
    if is_norm:
        if key.endswith("conv3.bn.weight"):
            # last BN of a residual branch: small gain keeps the synthetic trunk well conditioned
            # (activations O(1) through 14 blocks) like a trained net; otherwise they grow ~1.3x per block
            return ("uniform", 0.15, 0.45)
        return ("uniform", 0.5, 1.5) if last == "weight" else ("normal", 0.0, 0.1)
    mod = key.rsplit(".", 2)[-2] if key.count(".") >= 1 else ""
    mod_base = re.sub(r"\d+$", "", mod)
    if last == "bias":
        if mod_base in _DW_KEYS:
            return ("normal", 0.0, 0.02)
        return ("normal", 0.0, 0.1)
    # weights
    if mod_base in _DW_KEYS and len(shape) == 3 and shape[1] == 1:
        return ("normal", 0.0, 0.1)           # depthwise, reference init (modules.py:147-152)
    if mod.startswith("channel_conv"):
        return ("normal", 0.0, 1.0 / 3.0)
    fan_in = int(np.prod(shape[1:])) if len(shape) > 1 else int(shape[0])
    if "conv3D" in key or ".se." in key or "_fc_out" in key or "concat_fc" in key or ".mlp." in key:
        return ("normal", 0.0, float(1.0 / np.sqrt(fan_in)))
    return ("normal", 0.0, float(np.sqrt(2.0 / fan_in)))   # backbone convs


def make_state(shapes: dict, seed: int = 0) -> dict:
    """name -> numpy array for every entry of ``shapes`` (name -> (shape, dtype-string))."""
    out = {}
    for key, (shape, dt) in shapes.items():
        n = int(np.prod(shape)) if len(shape) else 1
        dist, a, b = _kind(key, shape)
        if dist == "zero":
            v = np.zeros(n)
        elif dist == "uniform":
            v = a + (b - a) * uniform01(seed, key, n)
        else:
            v = a + b * normalish(seed, key, n)
        out[key] = v.reshape(shape).astype(np.int64 if dt == "int64" else np.float32)
    return out
