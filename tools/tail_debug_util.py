"""Shared by tools/tail_debug.py and tools/tail_time.py: a random (attention, feed-forward) weight pair."""
import math

import torch


def pair_sd(C, H4, seed):
    g = torch.Generator().manual_seed(seed)

    def rn(*shape, s=1.0):
        return torch.randn(*shape, generator=g, dtype=torch.float64) * s
    return {"a.norm.gamma": 1 + 0.1 * rn(C), "a.to_qkv.weight": rn(3 * C, C, s=1.6 / math.sqrt(C)),
            "a.to_gates.weight": rn(C // 32, C, s=0.3), "a.to_gates.bias": rn(C // 32, s=0.3),
            "a.to_out.0.weight": rn(C, C, s=1 / math.sqrt(C)),
            "f.net.0.gamma": 1 + 0.1 * rn(C), "f.net.1.weight": rn(H4, C, s=1 / math.sqrt(C)),
            "f.net.1.bias": rn(H4, s=0.2), "f.net.4.weight": rn(C, H4, s=0.5 / math.sqrt(C)), "f.net.4.bias": rn(C, s=0.2)}
