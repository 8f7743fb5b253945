"""TEST INFRASTRUCTURE ONLY -- stand-in for rotary-embedding-torch==0.6.4.

The reference (beat_this/model/beat_tracker.py:11,52; roformer.py:121-123) imports
``RotaryEmbedding`` from a third-party wheel that is not installed in this image.
This module restates, from the published algorithm of that pinned version, exactly
the two things the reference touches: the ``freqs`` parameter (it appears in the
state_dict) and ``rotate_queries_or_keys``.  PARITY UNPINNED: the real wheel is not
available offline, so this restatement cannot be checked against it.
"""
import torch
from torch import nn


def _rotate_half(x):
    # interleaved pairs (x0,x1),(x2,x3)... -> (-x1,x0),(-x3,x2)...
    x = x.reshape(*x.shape[:-1], x.shape[-1] // 2, 2)
    x1, x2 = x.unbind(dim=-1)
    return torch.stack((-x2, x1), dim=-1).reshape(*x.shape[:-2], -1)


class RotaryEmbedding(nn.Module):
    def __init__(self, dim, theta=10000):
        super().__init__()
        freqs = 1.0 / (theta ** (torch.arange(0, dim, 2)[: dim // 2].float() / dim))
        self.freqs = nn.Parameter(freqs, requires_grad=False)

    def rotate_queries_or_keys(self, t, seq_dim=-2):
        n = t.shape[seq_dim]
        with torch.autocast(t.device.type, enabled=False):
            pos = torch.arange(n, device=t.device, dtype=torch.float32)
            ang = torch.einsum("n,f->nf", pos, self.freqs.float())
            ang = ang.repeat_interleave(2, dim=-1)  # [a0,a0,a1,a1,...]
            tf = t.float()
            out = tf * ang.cos() + _rotate_half(tf) * ang.sin()
        return out.to(t.dtype)
