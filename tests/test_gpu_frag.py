"""Parity of the fragment-major half-precision attention path on the MI355X (csrc/attn2.hip, csrc/qkv_front.hip)
against float64 torch restatements of roformer.py:83-132: attention on pre-arranged operands (all
sequence-length edge cases, the time-direction row scatter, the overflow fallback), and the frontend's
time-direction QKV projection producing those operands."""
import ctypes as Ct
import math

import pytest
import torch

from gpu_util import HALF, dev, frag_qk, frag_v, report, run_attn_frag, unfrag_qk, unfrag_v

pytestmark = pytest.mark.gpu


def _mk(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(shape, generator=g, dtype=torch.float64) * scale


def _rel(a, ref):
    return float((a.double().cpu() - ref).abs().max() / ref.abs().max().clamp_min(1e-30))


def _attn_ref(q, k, v, gates):
    s = q @ k.transpose(-1, -2) * math.log(2.0)  # q carries log2(e)/sqrt(d): softmax in base 2
    return torch.softmax(s, -1) @ v * gates[..., None]


def _is_f16():
    from beat_this_amd import _lib as Lb

    return not Lb.lib().bt_half_is_bf16()


def _run(q, k, v, gates, n_seq, L, heads, **omap):
    """q, k, v: [SH, L, 32] float64 (already representable in the half type); gates [SH, L]."""
    from beat_this_amd import _lib as Lb

    nbp = Lb.lib().bt_attn_frag_blocks(L)
    SH = n_seq * heads
    gh = torch.zeros((SH, nbp * 32), dtype=torch.float32)
    gh[:, :L] = gates.float()
    rows = n_seq * L
    out = torch.zeros((rows, heads * 32), dtype=HALF(), device=dev())
    # poison the padding blocks of K/V beyond ceil(L/32): they must never be consumed
    kf, vf = frag_qk(k.float(), nbp), frag_v(v.float(), nbp)
    nblk = (L + 31) // 32
    kf[:, nblk:] = float("nan")
    vf[:, nblk:] = float("nan")
    run_attn_frag(frag_qk(q.float(), nbp).to(dev()), kf.to(dev()), vf.to(dev()), gh.to(dev()), out, n_seq, L, heads, nbp,
                  **omap)
    return out


@pytest.mark.parametrize("variant", [-1, -2])
@pytest.mark.parametrize("n_seq,L,heads", [(3, 1500, 2), (2, 77, 1), (1, 128, 4), (5, 1012, 1), (2, 1, 1), (2, 33, 2),
                                           (1, 1499, 1), (2, 129, 1), (9, 96, 1), (2, 257, 1), (2, 250, 1), (1, 64, 1), (2, 65, 1)])
def test_attention_frag(n_seq, L, heads, variant):
    if variant == -2 and not _is_f16():
        pytest.skip("the two-query-block kernel is an fp16 kernel")
    SH = n_seq * heads
    q = _mk((SH, L, 32), 30, 0.6).float().to(HALF()).double()
    k = _mk((SH, L, 32), 31).float().to(HALF()).double()
    v = _mk((SH, L, 32), 32).float().to(HALF()).double()
    k[0, 7 % L] *= 6.0  # one outlier key
    k = k.float().to(HALF()).double()
    gates = torch.sigmoid(_mk((SH, L), 33))
    out = _run(q, k, v, gates, n_seq, L, heads, variant=variant)
    ref = _attn_ref(q, k, v, gates)  # [SH, L, 32]
    ref = ref.view(n_seq, heads, L, 32).permute(0, 2, 1, 3).reshape(n_seq * L, heads * 32)
    err = _rel(out, ref)
    report("attn_frag", n_seq=n_seq, L=L, heads=heads, variant=variant, rel=err)
    assert err < 2e-2


@pytest.mark.parametrize("n_seq,L,heads", [(3, 1500, 2), (2, 77, 1), (1, 128, 4), (5, 1012, 1), (2, 1, 1), (2, 33, 2), (1, 1499, 1),
                                           (2, 129, 1), (9, 96, 1), (2, 257, 1), (2, 250, 1), (1, 64, 1), (176, 1500, 1)])
def test_attention_frag_kernels_agree_bit_for_bit(n_seq, L, heads):
    """The two half attention kernels (128-query workgroups; two query blocks per wave on the hand-scheduled loop of round 6) must
    give the SAME bits for every query -- same reference point, same order of products, packing, row sums and P.V -- so that a
    launch-size-selected dispatch (a build switch: the forward runs the 128-query kernel, the faster one) could never make a
    chunk's logits depend on what else was in its batch.  Includes an outlier key (no overflow) and variant 0 (the dispatch rule)."""
    if not _is_f16():
        pytest.skip("the two-query-block kernel is an fp16 kernel")
    SH = n_seq * heads
    q = _mk((SH, L, 32), 30, 0.6).float().to(HALF()).double()
    k = _mk((SH, L, 32), 31).float().to(HALF()).double()
    v = _mk((SH, L, 32), 32).float().to(HALF()).double()
    k[0, 7 % L] *= 6.0
    k = k.float().to(HALF()).double()
    gates = torch.sigmoid(_mk((SH, L), 33))
    one = _run(q, k, v, gates, n_seq, L, heads, variant=-1)
    two = _run(q, k, v, gates, n_seq, L, heads, variant=-2)
    auto = _run(q, k, v, gates, n_seq, L, heads, variant=0)
    assert torch.equal(one, two) and torch.equal(one, auto)
    assert torch.equal(two, _run(q, k, v, gates, n_seq, L, heads, variant=-2))   # (repeatable)


def test_attention_frag_time_direction_rowmap():
    B, T, F, heads = 2, 150, 4, 1
    SH = B * F
    q, k, v = (_mk((SH, T, 32), 40 + i).float().to(HALF()).double() for i in range(3))
    gates = torch.sigmoid(_mk((SH, T), 44))
    out = _run(q, k, v, gates, SH, T, heads, o_div=F, o_outer=T * F, o_inner=1, o_tok=F)
    ref = _attn_ref(q, k, v, gates).view(B, F, T, 32).permute(0, 2, 1, 3).reshape(B * T * F, 32)
    err = _rel(out, ref)
    report("attn_frag_rowmap", rel=err)
    assert err < 2e-2


@pytest.mark.parametrize("variant", [-1, -2])
@pytest.mark.parametrize("L", [300, 1500])
def test_attention_frag_overflow_fallback(L, variant):
    """Scores that exceed the first key block's maximum by more than exp2 can hold force the SAFE
    (running-max) pass; the result must still be the exact softmax."""
    SH = 3
    q = _mk((SH, L, 32), 50, 0.5)
    k = _mk((SH, L, 32), 51)
    v = _mk((SH, L, 32), 52)
    # sequence 1: a late key aligned with query 5 with a huge score (~ +600 in log2 units)
    q[1, 5] = 0.0
    q[1, 5, 0] = 25.0
    k[1, L - 40] = 0.0
    k[1, L - 40, 0] = 24.0
    q, k, v = (t.float().to(HALF()).double() for t in (q, k, v))
    gates = torch.ones((SH, L), dtype=torch.float64)
    if variant == -2 and not _is_f16():
        pytest.skip("the two-query-block kernel is an fp16 kernel")
    out = _run(q, k, v, gates, SH, L, 1, variant=variant)
    ref = _attn_ref(q, k, v, gates).reshape(SH * L, 32)
    assert torch.isfinite(out.float()).all()
    err = _rel(out, ref)
    report("attn_frag_overflow", L=L, variant=variant, rel=err)
    assert err < 2e-2
    if variant == -2:
        # the repeat unit is the same 128 queries as in the one-block kernel (a wave pair here): identical bits with an overflow too,
        # wherever the overflowing query sits inside its 256-query workgroup
        assert torch.equal(out, _run(q, k, v, gates, SH, L, 1, variant=-1))
        q2 = q.clone()
        q2[1, 5] = q[1, 6]
        q2[1, 200 % L] = 0.0
        q2[1, 200 % L, 0] = 25.0
        q2 = q2.float().to(HALF()).double()
        assert torch.equal(_run(q2, k, v, gates, SH, L, 1, variant=-2), _run(q2, k, v, gates, SH, L, 1, variant=-1))


def _pair_sd(C, seed):
    H = C // 32
    g = torch.Generator().manual_seed(seed)

    def rn(*shape, s=1.0):
        return torch.randn(*shape, generator=g, dtype=torch.float64) * s
    return {
        "a.norm.gamma": 1 + 0.1 * rn(C), "a.to_qkv.weight": rn(3 * C, C, s=1.6 / math.sqrt(C)),
        "a.to_gates.weight": rn(H, C, s=0.3), "a.to_gates.bias": rn(H, s=0.3),
        "a.to_out.0.weight": rn(C, C, s=1 / math.sqrt(C)),
        "f.net.0.gamma": 1 + 0.1 * rn(C), "f.net.1.weight": rn(4 * C, C, s=1 / math.sqrt(C)),
        "f.net.1.bias": rn(4 * C, s=0.2), "f.net.4.weight": rn(C, 4 * C, s=0.5 / math.sqrt(C)),
        "f.net.4.bias": rn(C, s=0.2),
    }


@pytest.mark.parametrize("C", [32, 64, 128])
@pytest.mark.parametrize("T", [70, 1500])
def test_qkv_front(C, T):
    """RMSNorm + QKV + RoPE(time) + gates of the "(b f) t c" view, emitted fragment-major."""
    from beat_this_amd import _lib as L
    from beat_this_amd.pack import LOG2E, PackedPair
    from beat_this_amd.tables import rope_table

    H, F = C // 32, 1024 // C
    B = 2 if T < 1000 else 1
    if T >= 1000:
        F = 2  # keep the fp64 reference small; F is a runtime argument of the kernel
    sd = _pair_sd(C, 170 + C)
    x0 = _mk((B, T, F, C), 180 + C, 1.5)
    freqs = 10000.0 ** (-torch.arange(0, 32, 2).float() / 32)
    rope = torch.from_numpy(rope_table(freqs)).to(dev())
    pp = PackedPair(sd, "a.", "f.", C, dev())
    nbp = L.lib().bt_attn_frag_blocks(T)
    SH = B * F * H
    qf = torch.full((SH, nbp, 1024), float("nan"), dtype=HALF(), device=dev())
    kf, vf = qf.clone(), qf.clone()
    gh = torch.zeros((SH, nbp * 32), dtype=torch.float32, device=dev())
    xd = x0.float().to(dev())
    L.check(L.lib().bt_qkv_front(L.stream_ptr(dev()), L.PREC_HALF, Ct.byref(pp.weights), rope.data_ptr(), xd.data_ptr(), B, T, F,
                                 qf.data_ptr(), kf.data_ptr(), vf.data_ptr(), gh.data_ptr(), nbp))
    torch.cuda.synchronize()
    x = x0.float().double()
    xn = x / x.norm(dim=-1, keepdim=True).clamp_min(1e-12) * math.sqrt(C) * sd["a.norm.gamma"]
    qkv = (xn @ sd["a.to_qkv.weight"].T).reshape(B, T, F, 3, H, 32).permute(3, 0, 2, 4, 1, 5)  # qkv b f h t d
    ang = torch.arange(T, dtype=torch.float64)[:, None] * freqs.double()[None, :]
    cos, sin = ang.cos().repeat_interleave(2, -1), ang.sin().repeat_interleave(2, -1)

    def rot(t):
        te, to = t[..., 0::2], t[..., 1::2]
        return t * cos + torch.stack((-to, te), -1).flatten(-2) * sin
    q = rot(qkv[0]).reshape(SH, T, 32) * (LOG2E / math.sqrt(32.0))
    k = rot(qkv[1]).reshape(SH, T, 32)
    v = qkv[2].reshape(SH, T, 32)
    gates = torch.sigmoid(xn @ sd["a.to_gates.weight"].T + sd["a.to_gates.bias"])  # b t f h
    gates = gates.permute(0, 2, 3, 1).reshape(SH, T)
    nblk = (T + 31) // 32
    eq = _rel(unfrag_qk(qf.cpu()[:, :nblk], T), q)
    ek = _rel(unfrag_qk(kf.cpu()[:, :nblk], T), k)
    ev = _rel(unfrag_v(vf.cpu()[:, :nblk], T), v)
    eg = _rel(gh.cpu()[:, :T], gates)
    report("qkv_front", C=C, T=T, q=eq, k=ek, v=ev, gates=eg)
    assert max(eq, ek, ev) < 1.5e-2 and eg < 1e-2
    # tokens beyond T inside the last block are written as exact zeros for K and V (the attention kernel relies on it)
    if T % 32:
        tail_k = kf.cpu()[:, nblk - 1].view(SH, 4, 32, 8)[:, :, T % 32:, :]
        assert torch.all(tail_k.float() == 0)
        tail_v = unfrag_v(vf.cpu()[:, nblk - 1:nblk], 32)[:, T % 32:]
        assert torch.all(tail_v.float() == 0)


def _ff_ref(sd, x):
    C = x.shape[-1]
    xn = x / x.norm(dim=-1, keepdim=True).clamp_min(1e-12) * math.sqrt(C) * sd["f.net.0.gamma"]
    return x + torch.nn.functional.gelu(xn @ sd["f.net.1.weight"].T + sd["f.net.1.bias"]) @ sd["f.net.4.weight"].T \
        + sd["f.net.4.bias"]


@pytest.mark.parametrize("prec", [0, 1, 3])
@pytest.mark.parametrize("C", [32, 64, 128])
def test_fused_out_ff(prec, C):
    """x += to_out(ao); x += FF(x) in one launch (csrc/fused2.hip), both precisions, ragged M."""
    from beat_this_amd import _lib as L
    from beat_this_amd.pack import PackedPair

    sd = _pair_sd(C, 250 + C)
    M = 1000 + C
    x0 = _mk((M, C), 260 + C, 1.5)
    dt = torch.float32 if prec != 1 else HALF()   # (prec 3 = BT_PREC_F32X3: fp32 in memory, hi + lo half operands)
    ao = _mk((M, C), 270 + C).float().to(dt)
    pp = PackedPair(sd, "a.", "f.", C, dev())
    x = x0.float().to(dev()).clone()
    aod = ao.to(dev())
    L.check(L.lib().bt_outff_fused(L.stream_ptr(dev()), prec, Ct.byref(pp.weights), aod.data_ptr(), x.data_ptr(), M, 0))
    torch.cuda.synchronize()
    x1 = x0.float().double() + ao.double() @ sd["a.to_out.0.weight"].T
    ref = _ff_ref(sd, x1)
    err = _rel(x, ref)
    report("outff_fused", prec=prec, C=C, rel=err)
    assert err < (2e-5 if prec != 1 else 1.5e-2)


@pytest.mark.parametrize("prec", [0, 1, 3])
@pytest.mark.parametrize("C", [32, 64, 128])
def test_fused_attn_ff(prec, C):
    """x += AttnF(x); x += FF(x) over the F = 1024/C tokens of each (b,t) row in one launch."""
    from beat_this_amd import _lib as L
    from beat_this_amd.pack import PackedPair
    from beat_this_amd.tables import rope_table

    H, F = C // 32, 1024 // C
    sd = _pair_sd(C, 370 + C)
    rows = 37
    M = rows * F
    x0 = _mk((M, C), 380 + C, 1.5)
    freqs = 10000.0 ** (-torch.arange(0, 32, 2).float() / 32)
    rope = torch.from_numpy(rope_table(freqs)).to(dev())
    pp = PackedPair(sd, "a.", "f.", C, dev())
    x = x0.float().to(dev()).clone()
    L.check(L.lib().bt_attnff_fused(L.stream_ptr(dev()), prec, Ct.byref(pp.weights), rope.data_ptr(), x.data_ptr(), M))
    torch.cuda.synchronize()
    xx = x0.float().double()
    xn = xx / xx.norm(dim=-1, keepdim=True).clamp_min(1e-12) * math.sqrt(C) * sd["a.norm.gamma"]
    qkv = (xn @ sd["a.to_qkv.weight"].T).reshape(rows, F, 3, H, 32).permute(2, 0, 3, 1, 4)
    ang = torch.arange(F, dtype=torch.float64)[:, None] * freqs.double()[None, :]
    cos, sin = ang.cos().repeat_interleave(2, -1), ang.sin().repeat_interleave(2, -1)

    def rot(t):
        te, to = t[..., 0::2], t[..., 1::2]
        return t * cos + torch.stack((-to, te), -1).flatten(-2) * sin
    q, k, v = rot(qkv[0]), rot(qkv[1]), qkv[2]
    att = torch.softmax(q @ k.transpose(-1, -2) / math.sqrt(32.0), -1) @ v
    gates = torch.sigmoid(xn @ sd["a.to_gates.weight"].T + sd["a.to_gates.bias"]).reshape(rows, F, H).permute(0, 2, 1)
    out = (att * gates[..., None]).permute(0, 2, 1, 3).reshape(M, C) @ sd["a.to_out.0.weight"].T
    ref = _ff_ref(sd, xx + out)
    err = _rel(x, ref)
    report("attnff_fused", prec=prec, C=C, rel=err)
    assert err < (3e-5 if prec != 1 else 2e-2)


@pytest.mark.parametrize("prec", [0, 1, 3])
@pytest.mark.parametrize("C", [32, 128])
def test_fused_halves_at_scale_are_repeatable(C, prec):
    """Many workgroups per CU (model scale): four launches of each fused half on identical inputs must agree bit for bit
    (the weight ring's write-after-read race -- raw s_barrier without lgkmcnt(0) -- corrupted a few 32-token blocks per
    launch only at this scale), and the two precisions must agree to half-operand accuracy."""
    from beat_this_amd import _lib as L
    from beat_this_amd.pack import PackedPair
    from beat_this_amd.tables import rope_table

    sd = _pair_sd(C, 5 + C)
    pp = PackedPair(sd, "a.", "f.", C, dev())
    M = 16 * 1500 * 1024 // C
    rope = torch.from_numpy(rope_table(10000.0 ** (-torch.arange(0, 32, 2).float() / 32))).to(dev())
    x0 = _mk((M, C), 7 + C, 1.5).float().to(dev())
    ao = _mk((M, C), 9 + C).float().to(torch.float32 if prec != 1 else HALF()).to(dev())
    st = L.stream_ptr(dev())
    outs_a, outs_o = [], []
    for _ in range(4):
        xa, xo = x0.clone(), x0.clone()
        L.check(L.lib().bt_attnff_fused(st, prec, Ct.byref(pp.weights), rope.data_ptr(), xa.data_ptr(), M))
        L.check(L.lib().bt_outff_fused(st, prec, Ct.byref(pp.weights), ao.data_ptr(), xo.data_ptr(), M, 0))
        outs_a.append(xa)
        outs_o.append(xo)
    torch.cuda.synchronize()
    bad = sum(int(not torch.equal(o, outs_a[0])) for o in outs_a[1:]) + sum(int(not torch.equal(o, outs_o[0])) for o in outs_o[1:])
    report("fused2_scale_repeatable", C=C, prec=prec, deviating=bad)
    assert bad == 0


@pytest.mark.parametrize("C,M", [(512, 777), (512, 4096 + 33), (512, 24000), (256, 777), (256, 24000)])
def test_layer_tail(C, M):
    """Fused tail of a main layer (csrc/tail.hip): x += to_out(ao); x += FF(x), the half shadow and the per-64-column
    partial sums of squares of the new x, against fp64 (exact operands: the tolerance covers half operand rounding)."""
    from beat_this_amd import _lib as L
    from beat_this_amd.pack import PackedPair

    sd = _pair_sd(C, 450 + C)
    x0 = _mk((M, C), 460 + C, 1.5)
    ao = _mk((M, C), 470 + C).float().to(HALF())
    pp = PackedPair(sd, "a.", "f.", C, dev())
    assert pp.weights.w_tail_frag
    x = x0.float().to(dev()).clone()
    aod = ao.to(dev())
    xb = torch.full((M, C), float("nan"), dtype=HALF(), device=dev())
    ssq = torch.full((C // 64, M), float("nan"), dtype=torch.float32, device=dev())
    L.check(L.lib().bt_layer_tail(L.stream_ptr(dev()), Ct.byref(pp.weights), 4 * C, aod.data_ptr(), x.data_ptr(), M,
                                  xb.data_ptr(), ssq.data_ptr()))
    torch.cuda.synchronize()
    x1 = x0.float().double() + ao.double() @ sd["a.to_out.0.weight"].T
    ref = _ff_ref(sd, x1)
    err = _rel(x, ref)
    xc = x.double().cpu()
    err_b = float((xb.double().cpu() - xc).abs().max() / xc.abs().max())
    ref_ssq = (xc * xc).view(M, C // 64, 64).sum(-1).T
    err_s = float((ssq.double().cpu() - ref_ssq).abs().max() / ref_ssq.abs().max())
    # a second launch on the same inputs must reproduce the result bit for bit (no atomics, fixed accumulation order)
    x2 = x0.float().to(dev()).clone()
    L.check(L.lib().bt_layer_tail(L.stream_ptr(dev()), Ct.byref(pp.weights), 4 * C, aod.data_ptr(), x2.data_ptr(), M, 0, 0))
    torch.cuda.synchronize()
    report("layer_tail", C=C, M=M, rel=err, rel_shadow=err_b, rel_ssq=err_s)
    assert torch.equal(x, x2)
    assert err < (3e-3 if HALF() == torch.float16 else 1.5e-2)
    assert err_b < 2e-3 if HALF() == torch.float16 else err_b < 1e-2
    assert err_s < 1e-5
