"""BASELINE config 5 at operator level (report-only): the MX e4m3 GEMM of csrc/gemm_mx8.hip against float64 on the SAME quantised
operands (the kernel's job is the product of what it is given: block scales applied per 32 k, fp32 accumulation), at the main
layers' GEMM shapes (roformer.py:38-61,99-132).  The arithmetic's cost in logits / beats is priced on the oracle
(tools/flip_soak.py sim --schemes mxfp8,halfsim -> profiles/r06_cfg5_mx8.txt)."""
import numpy as np
import pytest
import torch

from gpu_util import dev, report

pytestmark = pytest.mark.gpu


def mx_quantise(x: torch.Tensor):
    """fp32 [R, K] -> (e4m3 bytes uint8 [R, K], E8M0 scale bytes uint8 [R, K / 32], the dequantised float64 values): OCP MX, the
    block maximum lands in e4m3's top binade (scale = 2^(floor(log2 amax) - 8))"""
    R, K = x.shape
    xb = x.double().view(R, K // 32, 32)
    amax = xb.abs().amax(-1, keepdim=True).clamp_min(2.0 ** -100)
    e = (torch.floor(torch.log2(amax)) - 8).clamp(-127, 127)
    q = (xb / torch.exp2(e)).clamp(-448, 448).float().to(torch.float8_e4m3fn)
    deq = (q.float().double() * torch.exp2(e)).view(R, K)
    return q.view(torch.uint8).view(R, K).contiguous(), (e + 127).to(torch.uint8).view(R, K // 32).contiguous(), deq


@pytest.mark.parametrize("M,K,N", [(1500, 512, 2048), (24000, 512, 2048), (3000, 2048, 512), (333, 512, 512), (4100, 1024, 1536),
                                   (24000, 2048, 512), (130, 512, 64)])
def test_gemm_mx8_matches_float64_on_the_quantised_operands(M, K, N):
    from beat_this_amd import _lib as L

    if not hasattr(torch, "float8_e4m3fn"):
        pytest.skip("torch without float8_e4m3fn")
    g = torch.Generator().manual_seed(M + K + N)
    # activations with a few large channels and weights of very different row magnitudes: the block scales have to do real work
    a = torch.randn((M, K), generator=g) * (1.0 + 30.0 * (torch.rand((1, K), generator=g) > 0.97))
    w = torch.randn((N, K), generator=g) * torch.exp2(torch.randint(-9, 3, (N, 1), generator=g).float())
    npad = (N + 127) // 128 * 128
    wq = torch.zeros((npad, K))
    wq[:N] = w
    ab, asc, adeq = mx_quantise(a)
    wb, wsc, wdeq = mx_quantise(wq)
    out = torch.empty((M, N), dtype=torch.float32, device=dev())
    d = [t.to(dev()) for t in (ab, asc, wb, wsc)]
    L.check(L.lib().bt_gemm_mx8(L.stream_ptr(dev()), d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), d[3].data_ptr(),
                                out.data_ptr(), M, N, K, N))
    torch.cuda.synchronize()
    ref = adeq @ wdeq[:N].T
    err = float((out.double().cpu() - ref).abs().max() / ref.abs().max())
    # what the format itself costs against the unquantised product (reported: this is config 5's arithmetic, not the kernel's error)
    fmt = float((ref - a.double() @ w.double().T).abs().max() / ref.abs().max())
    report("gemm_mx8", M=M, K=K, N=N, rel_vs_quantised_f64=err, rel_format_vs_unquantised=fmt)
    # (measured: 1.2e-5 .. 2.1e-5 on Gaussian operands -- unit scales included -- and 6e-5 .. 1.4e-4 with the outlier channels of this
    # test: the f8f6f4 MFMA sums its 64 exact products per issue with far fewer guard bits than the 2^-24 the fp16 / fp32 MFMAs
    # deliver, relative to the largest term of the group -- tools/mx8_debug.py; a property of the instruction, priced into config 5)
    assert err < 3e-4, err
    assert fmt < 0.2
