"""Parity of the main-layer half-precision GEMM (csrc/gemm3.hip) on the MI355X against float64 torch restatements:
FF1 (RMSNorm factor from partial sums of squares + bias + GELU), residual update (+ half shadow + partial sums
of squares for the next RMSNorm), and the QKV projection with RoPE / gates written fragment-major."""
import ctypes as C
import math

import pytest
import torch

from gpu_util import HALF, dev, pad_rows, report, unfrag_qk, unfrag_v

pytestmark = pytest.mark.gpu


def _mk(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(shape, generator=g, dtype=torch.float64) * scale


def _rel(a, ref):
    return float((a.double().cpu() - ref).abs().max() / ref.abs().max().clamp_min(1e-30))


def _ssq_parts(x32):
    """[M, D] fp32 -> [D/64][M] partial sums of squares (what the producers of the residual stream emit)."""
    M, D = x32.shape
    return (x32.double() ** 2).view(M, D // 64, 64).sum(-1).T.contiguous().float()


def _call(**kw):
    from beat_this_amd import _lib as L

    a = L.Gemm3Args()
    for k, v in kw.items():
        setattr(a, k, v.data_ptr() if isinstance(v, torch.Tensor) else v)
    L.check(L.lib().bt_gemm3(L.stream_ptr(dev()), C.byref(a)))
    torch.cuda.synchronize()


@pytest.mark.parametrize("M,K,N", [(1500, 512, 2048), (333, 128, 512), (24000, 512, 2048)])
def test_gemm3_ff1(M, K, N):
    x = _mk((M, K), 1, 2.0).float()
    W, b = _mk((N, K), 2, 1 / math.sqrt(K)), _mk((N,), 3)
    xb = x.to(HALF())
    Wd = W.float().to(HALF())
    out = torch.zeros((M, N), dtype=HALF(), device=dev())
    ssq = _ssq_parts(x)
    _call(A=xb.to(dev()), lda=K, M=M, K=K, W=Wd.to(dev()), N=N, epi=0, bias=b.float().to(dev()), ssq_in=ssq.to(dev()),
          ssq_parts=K // 64, out=out, ldo=N)
    rs = math.sqrt(K) / x.double().norm(dim=-1, keepdim=True).clamp_min(1e-12)
    ref = torch.nn.functional.gelu(xb.double() @ Wd.double().T * rs + b)
    err = _rel(out, ref)
    report("gemm3_ff1", M=M, K=K, N=N, rel=err)
    assert err < 6e-3


# (M = 24000 runs on the 192 x 256 tiles, M = 32768 on the 256 x 256 ones, the others on 128 x 128)
@pytest.mark.parametrize("M,K,N,bias", [(1500, 2048, 512, True), (777, 512, 512, False), (130, 128, 128, True),
                                        (24000, 2048, 512, True), (32768, 1024, 512, False)])
def test_gemm3_resid(M, K, N, bias):
    A = _mk((M, K), 4).float().to(HALF())
    W = _mk((N, K), 5, 0.5 / math.sqrt(K)).float().to(HALF())
    b = _mk((N,), 6)
    x0 = _mk((M, N), 7).float()
    x = x0.to(dev()).clone()
    xb = torch.zeros((M, N), dtype=HALF(), device=dev())
    ssq = torch.full((N // 64, M), -1.0, dtype=torch.float32, device=dev())
    _call(A=A.to(dev()), lda=K, M=M, K=K, W=W.to(dev()), N=N, epi=1, bias=b.float().to(dev()) if bias else 0, x=x, ldx=N,
          xb=xb, ssq_out=ssq)
    ref = x0.double() + A.double() @ W.double().T + (b if bias else 0)
    err = _rel(x, ref)
    errb = _rel(xb, ref)
    ssq_ref = (ref ** 2).view(M, N // 64, 64).sum(-1).T
    errs = _rel(ssq, ssq_ref)
    report("gemm3_resid", M=M, K=K, N=N, rel=err, shadow=errb, ssq=errs)
    assert err < 1e-5 and errb < 5e-3 and errs < 1e-4


@pytest.mark.parametrize("n_seq,L,heads", [(2, 1500, 4), (3, 77, 4), (1, 1, 4), (5, 130, 8)])
def test_gemm3_qkv(n_seq, L, heads):
    from beat_this_amd import _lib as Lb
    from beat_this_amd.pack import LOG2E
    from beat_this_amd.tables import rope_table

    D = heads * 32
    M = n_seq * L
    x = _mk((M, D), 10, 1.5).float()
    Wqkv = _mk((3 * D, D), 11, 1.6 / math.sqrt(D))
    Wqkv[:D] *= LOG2E / math.sqrt(32.0)
    Wg, bg = _mk((heads, D), 12, 0.3), _mk((heads,), 13, 0.3)
    W = pad_rows(torch.cat([Wqkv, Wg]).float()).to(HALF())
    freqs = 10000.0 ** (-torch.arange(0, 32, 2).float() / 32)
    rope = torch.from_numpy(rope_table(freqs)).to(dev())
    nbp = Lb.lib().bt_attn_frag_blocks(L)
    SH = n_seq * heads
    qf = torch.full((SH, nbp, 1024), float("nan"), dtype=HALF(), device=dev())
    kf, vf = qf.clone(), qf.clone()
    gh = torch.zeros((SH, nbp * 32), dtype=torch.float32, device=dev())
    xb = x.to(HALF())
    _call(A=xb.to(dev()), lda=D, M=M, K=D, W=W.to(dev()), N=3 * D + heads, epi=2, ssq_in=_ssq_parts(x).to(dev()),
          ssq_parts=D // 64, n_seq=n_seq, L=L, nbp=nbp, heads=heads, rope=rope, qf=qf, kf=kf, vf=vf, gates=gh,
          b_gates=bg.float().to(dev()))
    rs = math.sqrt(D) / x.double().norm(dim=-1, keepdim=True).clamp_min(1e-12)
    Wd = W.double()
    qkv = (xb.double() @ Wd[:3 * D].T * rs).view(n_seq, L, 3, heads, 32).permute(2, 0, 3, 1, 4)  # qkv s h t d
    ang = torch.arange(L, dtype=torch.float64)[:, None] * freqs.double()[None, :]
    cos, sin = ang.cos().repeat_interleave(2, -1), ang.sin().repeat_interleave(2, -1)

    def rot(t):
        te, to = t[..., 0::2], t[..., 1::2]
        return t * cos + torch.stack((-to, te), -1).flatten(-2) * sin
    q, k, v = rot(qkv[0]).reshape(SH, L, 32), rot(qkv[1]).reshape(SH, L, 32), qkv[2].reshape(SH, L, 32)
    gates = torch.sigmoid(xb.double() @ Wd[3 * D:3 * D + heads].T * rs + bg).view(n_seq, L, heads).permute(0, 2, 1).reshape(SH, L)
    nblk = (L + 31) // 32
    eq = _rel(unfrag_qk(qf.cpu()[:, :nblk], L), q)
    ek = _rel(unfrag_qk(kf.cpu()[:, :nblk], L), k)
    ev = _rel(unfrag_v(vf.cpu()[:, :nblk], L), v)
    eg = _rel(gh.cpu()[:, :L], gates)
    report("gemm3_qkv", n_seq=n_seq, L=L, heads=heads, q=eq, k=ek, v=ev, gates=eg)
    assert max(eq, ek, ev) < 6e-3 and eg < 2e-3
    if L % 32:  # tokens beyond L inside the last block: exact zeros for K and V
        tail_k = kf.cpu()[:, nblk - 1].view(SH, 4, 32, 8)[:, :, L % 32:, :]
        assert torch.all(tail_k.float() == 0)
        assert torch.all(unfrag_v(vf.cpu()[:, nblk - 1:nblk], 32)[:, L % 32:].float() == 0)


def test_gemm3_store_without_residual():
    """epi 1 with no_resid (frontend.linear): x is written, never read (NaN-filled on entry)."""
    M, K, N = 3000, 1024, 512
    A = _mk((M, K), 51).float().to(HALF())
    W = _mk((N, K), 52, 1 / math.sqrt(K)).float().to(HALF())
    b = _mk((N,), 53)
    x = torch.full((M, N), float("nan"), dtype=torch.float32, device=dev())
    xb = torch.zeros((M, N), dtype=HALF(), device=dev())
    ssq = torch.zeros((N // 64, M), dtype=torch.float32, device=dev())
    _call(A=A.to(dev()), lda=K, M=M, K=K, W=W.to(dev()), N=N, epi=1, bias=b.float().to(dev()), x=x, ldx=N, xb=xb,
          ssq_out=ssq, no_resid=1)
    ref = A.double() @ W.double().T + b
    err, errb = _rel(x, ref), _rel(xb, ref)
    errs = _rel(ssq, (ref ** 2).view(M, N // 64, 64).sum(-1).T)
    report("gemm3_store", M=M, K=K, N=N, rel=err, shadow=errb, ssq=errs)
    assert err < 1e-5 and errb < 5e-3 and errs < 1e-4


@pytest.mark.parametrize("B,T,Fp,C2,N,half_out", [(2, 50, 8, 128, 128, False), (1, 33, 4, 256, 256, True), (3, 7, 16, 64, 128, False),
                                                   (2, 40, 16, 64, 64, False)])
def test_gemm3_frontend_conv(B, T, Fp, C2, N, half_out):
    """epi 1 as the (2,3) / stride (2,1) frontend convolution on the half (b, t, f, c) activation: three time taps gathered
    by the LDS-DMA loader (zero rows outside 0 <= t < T), bias, tanh-form GELU; fp32 or half output."""
    M, K = B * T * Fp, 3 * C2
    x = _mk((M, C2), 71).float().to(HALF())               # row m = (b, t, f'), C2 = 2 C channels of the frequency pair
    W = _mk((N, K), 72, 1 / math.sqrt(K)).float().to(HALF())
    b = _mk((N,), 73, 0.5)
    out = torch.full((M, N), float("nan"), dtype=torch.float32, device=dev())
    outb = torch.zeros((M, N), dtype=HALF(), device=dev())
    _call(A=x.to(dev()), lda=C2, M=M, K=K, W=pad_rows(W).to(dev()), N=N, epi=1, bias=b.float().to(dev()),
          x=0 if half_out else out, ldx=N, xb=outb if half_out else 0, no_resid=1, gelu=1, conv_C2=C2, conv_T=T, conv_F=Fp)
    xd = x.double().view(B, T, Fp, C2)
    xp = torch.zeros((B, T + 2, Fp, C2), dtype=torch.float64)
    xp[:, 1:-1] = xd
    Wd = W.double()
    acc = sum(xp[:, dt:dt + T].reshape(M, C2) @ Wd[:, dt * C2:(dt + 1) * C2].T for dt in range(3))
    ref = torch.nn.functional.gelu(acc + b)  # exact GELU: the kernel's fitted sigmoid form is within 2.6e-5 of it (common.h)
    err = _rel(outb if half_out else out, ref)
    report("gemm3_frontend_conv", B=B, T=T, Fp=Fp, C2=C2, N=N, rel=err)
    assert err < (5e-3 if half_out else 4e-5)
