"""state_dict key grammar of ``TDEEDModel.Impl``, rebuilt from the config.

Keys, shapes and ORDER equal what ``self._model.state_dict()`` yields in the reference
(model/model.py:25-103 + timm RegNet + model/shift.py + model/impl/gsf.py + model/modules.py),
so reference checkpoints load here and ours load there.  ``tests/test_oracle_golden.py``
pins this against a digest recorded from the reference model itself.
"""
import hashlib
from collections import OrderedDict

from .regnet_spec import regnet_spec, sgp_up_size

F32, I64 = "float32", "int64"


def _bn(d, pre, c):
    d[pre + ".weight"] = ((c,), F32)
    d[pre + ".bias"] = ((c,), F32)
    d[pre + ".running_mean"] = ((c,), F32)
    d[pre + ".running_var"] = ((c,), F32)
    d[pre + ".num_batches_tracked"] = ((), I64)


def _conv_bn(d, pre, cin, cout, k, groups=1):
    d[pre + ".conv.weight"] = ((cout, cin // groups, k, k), F32)
    _bn(d, pre + ".bn", cout)


def _wb(d, pre, wshape):
    d[pre + ".weight"] = (tuple(wshape), F32)
    d[pre + ".bias"] = ((wshape[0],), F32)


def _gate_shift(d, pre, fold, mode):
    _wb(d, pre + ".conv3D", (2, fold // 2, 3, 3, 3))
    _bn(d, pre + ".bn", fold)
    if mode == "gsf":
        _wb(d, pre + ".channel_conv1", (1, 2, 3, 3))
        _wb(d, pre + ".channel_conv2", (1, 2, 3, 3))


def _sgp_block(d, pre, C, ks, up):
    d[pre + ".ln.weight"] = ((1, C, 1), F32)
    d[pre + ".ln.bias"] = ((1, C, 1), F32)
    _wb(d, pre + ".gn", (C,))
    _wb(d, pre + ".psi", (C, 1, ks))
    _wb(d, pre + ".fc", (C, 1, 1))
    _wb(d, pre + ".convw", (C, 1, ks))
    _wb(d, pre + ".convkw", (C, 1, up))
    _wb(d, pre + ".global_fc", (C, 1, 1))
    _wb(d, pre + ".mlp.0", (4 * C, C, 1))
    _wb(d, pre + ".mlp.2", (C, 4 * C, 1))


def _sgp_mixer(d, pre, C, ks, up):
    for n in ("ln1", "ln2"):
        d[f"{pre}.{n}.weight"] = ((1, C, 1), F32)
        d[f"{pre}.{n}.bias"] = ((1, C, 1), F32)
    _wb(d, pre + ".gn", (C,))
    _wb(d, pre + ".psi1", (C, 1, ks))
    _wb(d, pre + ".psi2", (C, 1, ks))
    _wb(d, pre + ".convw1", (C, 1, ks))
    _wb(d, pre + ".convkw1", (C, 1, up))
    _wb(d, pre + ".convw2", (C, 1, ks))
    _wb(d, pre + ".convkw2", (C, 1, up))
    _wb(d, pre + ".fc1", (C, 1, 1))
    _wb(d, pre + ".global_fc1", (C, 1, 1))
    _wb(d, pre + ".fc2", (C, 1, 1))
    _wb(d, pre + ".global_fc2", (C, 1, 1))
    _wb(d, pre + ".mlp.0", (4 * C, C, 1))
    _wb(d, pre + ".mlp.2", (C, 4 * C, 1))
    _wb(d, pre + ".concat_fc", (C, 6 * C, 1))


def model_state_shapes(cfg, double_head=None) -> "OrderedDict[str, tuple]":
    """cfg: mapping/namespace with feature_arch, clip_len, n_layers, sgp_ks, sgp_r, num_classes,
    radi_displacement.  double_head: None or [k1, k2] (``update_pred_head``, model/model.py:169-172)."""
    g = (lambda k: cfg[k]) if isinstance(cfg, dict) else (lambda k: getattr(cfg, k))
    arch = g("feature_arch")
    spec = regnet_spec(arch)
    mode = "gsm" if arch.endswith("_gsm") else ("gsf" if arch.endswith("_gsf") else None)
    C = spec.feat_dim
    T = g("clip_len")
    d = OrderedDict()
    d["temp_enc"] = ((T, C), F32)
    _conv_bn(d, "_features.stem", 3, 32, 3)
    for b in spec.blocks:
        p = "_features." + b.name
        if b.gsf_fold:
            _gate_shift(d, p + ".conv1.gs", b.gsf_fold, mode)
            _conv_bn(d, p + ".conv1.net", b.cin, b.cout, 1)
        else:
            _conv_bn(d, p + ".conv1", b.cin, b.cout, 1)
        _conv_bn(d, p + ".conv2", b.cout, b.cout, 3, groups=b.groups)
        _wb(d, p + ".se.fc1", (b.se_rd, b.cout, 1, 1))
        _wb(d, p + ".se.fc2", (b.cout, b.se_rd, 1, 1))
        _conv_bn(d, p + ".conv3", b.cout, b.cout, 1)
        if b.has_downsample:
            _conv_bn(d, p + ".downsample", b.cin, b.cout, 1)
    n, ks = g("n_layers"), g("sgp_ks")
    up = sgp_up_size(ks, g("sgp_r"))
    for i in range(2 * n + 1):
        _sgp_block(d, f"_temp_fine._sgp.{i}", C, ks, up)
    for i in range(n):
        _sgp_mixer(d, f"_temp_fine._sgpMixer.{i}", C, ks, up)
    if double_head:
        _wb(d, "_pred_fine._fc1._fc_out", (double_head[0], C))
        _wb(d, "_pred_fine._fc2._fc_out", (double_head[1], C))
    else:
        _wb(d, "_pred_fine._fc_out", (g("num_classes") + 1, C))
    if g("radi_displacement") > 0:
        _wb(d, "_pred_displ._fc_out", (1, C))
    return d


def layout_digest(shapes) -> str:
    s = "\n".join(f"{k}:{tuple(v[0])}" for k, v in shapes.items())
    return hashlib.sha1(s.encode()).hexdigest()


def is_parameter(key: str) -> bool:
    last = key.rsplit(".", 1)[-1]
    return last not in ("running_mean", "running_var", "num_batches_tracked")
