"""Counterparts of the helper functions in the reference's ``model/modules.py`` (406-438)."""
import torch

from . import ops


def process_prediction(pred, predD):
    """modules.py:406-414 on the GPU: softmax + displacement-guided scatter-max, one launch,
    no per-frame host sync.  pred (B,T,K1) fp32, predD (B,T) fp32 -> (B,T,K1)."""
    B, T, K1 = pred.shape
    head = torch.cat([pred.float(), predD.float().unsqueeze(-1)], dim=-1).reshape(B * T, K1 + 1).contiguous()
    _, scores = ops.process_prediction(head, B, T, K1, K1)
    return scores


def process_double_head(pred, predD, num_classes=1):
    """modules.py:416-426: only the first head's classes take part."""
    return process_prediction(pred[:, :, :num_classes].contiguous(), predD)


def process_labels(label, labelD, num_classes=18):
    """modules.py:428-438 (host-side label preparation for mAP; KB-sized, stays on the CPU)."""
    label = torch.as_tensor(label)
    B, T = label.shape
    out = torch.zeros((B, T, num_classes))
    out[:, :, 0] = 1
    ev = label.nonzero()
    if ev.numel() == 0:
        return out
    b, t = ev[:, 0], ev[:, 1]
    d = torch.as_tensor(labelD)[b, t].to(torch.int64) if labelD is not None else torch.zeros_like(t)
    tt = t - d
    ok = (tt >= 0) & (tt < T)
    b, tt, c = b[ok], tt[ok], label[b, t][ok]
    out[b, tt, c] = 1
    out[b, tt, 0] = 0
    return out
