"""The callable sub-modules below the three stages (model.frontend.stem / .blocks / .blocks[i] / .blocks[i].partial / .concat /
.linear, model.transformer_blocks.layers[l][0] / [l][1] / .norm; reference: beat_tracker.py:54-80,108-168,
roformer.py:138-181) against the oracle's functions of the same names, in the reference's tensor layouts."""
import pytest
import torch

from gpu_util import dev, report

pytestmark = pytest.mark.gpu
HP = dict(transformer_dim=256, n_layers=2)   # (small model: the sub-module entry runs the generic kernels)


def _model(style="lively", **hp):
    from beat_this_amd import weights as W
    from beat_this_amd.model import BeatThis

    h = W.resolve_hparams(dict(HP, **hp))
    sd = W.random_state_dict(h, seed=5, style=style)
    m = BeatThis(**h)
    m.load_state_dict(sd)
    # (sub-module calls run the exact fp32 kernels; the module's own default for whole stages and forwards is the hi + lo path
    # since round 5 -- these tests compare like with like, the last one also checks the default against the hooked chain)
    m.fp32_split_gemms = False
    return m.to(dev()).eval(), {k: v.double() if v.is_floating_point() else v for k, v in sd.items()}, h


def _rel(a, ref):
    return float((a.double().cpu() - ref).abs().max() / ref.abs().max().clamp_min(1e-30))


def test_frontend_submodules_match_the_oracle():
    from beat_this_amd import weights as W
    from oracle import beat_this_oracle as O

    m, sd, _ = _model()
    x = torch.from_numpy(W.synthetic_spect(200, seed=2))[None].repeat(2, 1, 1)
    x[1] = x[1].flip(0)
    with torch.inference_mode():
        s_ref = O.stem(x.double(), sd)
        s = m.frontend.stem(x.to(dev()))
        assert s.shape == s_ref.shape == (2, 32, 32, 200)
        errs = {"stem": _rel(s, s_ref)}
        cur_ref, cur = s_ref, s
        for i in range(3):
            p = f"frontend.blocks.{i}."
            pr = O.partial_ft(cur_ref, sd, p + "partial.")
            errs[f"partial{i}"] = _rel(m.frontend.blocks[i].partial(cur_ref.float().to(dev())), pr)
            br = torch.nn.functional.gelu(O.batchnorm(torch.nn.functional.conv2d(pr, sd[p + "conv2d.weight"], stride=(2, 1), padding=(0, 1)),
                                                      sd, p + "norm.", 1))
            errs[f"block{i}"] = _rel(m.frontend.blocks[i](cur_ref.float().to(dev())), br)
            cur_ref = br
        errs["blocks"] = _rel(m.frontend.blocks(s_ref.float().to(dev())), cur_ref)
        cat_ref = cur_ref.permute(0, 3, 1, 2).reshape(2, 200, 1024)
        cat = m.frontend.concat(cur_ref.float().to(dev()))
        assert torch.equal(cat.cpu().double(), cat_ref.float().double())
        lin_ref = cat_ref @ sd["frontend.linear.weight"].T + sd["frontend.linear.bias"]
        errs["linear"] = _rel(m.frontend.linear(cat), lin_ref)
        # composed like the reference's Sequential: equals the frontend stage
        whole = m.frontend.linear(m.frontend.concat(m.frontend.blocks(m.frontend.stem(x.to(dev())))))
        errs["composed_vs_stage"] = _rel(whole, m.frontend(x.to(dev())).double().cpu())
    report("submodules_frontend", **errs)
    assert max(errs.values()) < 2e-5, errs


def test_transformer_submodules_match_the_oracle():
    from oracle import beat_this_oracle as O

    m, sd, h = _model()
    D = h["transformer_dim"]
    g = torch.Generator().manual_seed(3)
    x = torch.randn((2, 333, D), generator=g)
    errs = {}
    with torch.inference_mode():
        for l in range(h["n_layers"]):
            p = f"transformer_blocks.layers.{l}."
            a_ref = O.attention(x.double(), sd, p + "0.", D // 32)
            f_ref = O.feedforward(x.double(), sd, p + "1.")
            errs[f"attn{l}"] = _rel(m.transformer_blocks.layers[l][0](x.to(dev())), a_ref)
            errs[f"ff{l}"] = _rel(m.transformer_blocks.layers[l][1](x.to(dev())), f_ref)
        errs["norm"] = _rel(m.transformer_blocks.norm(x.to(dev())), O.rmsnorm(x.double(), sd["transformer_blocks.norm.gamma"]))
        # the reference's loop over the ModuleList, written against this model
        y = x.to(dev())
        for attn, ff in m.transformer_blocks.layers:
            y = attn(y) + y
            y = ff(y) + y
        y = m.transformer_blocks.norm(y)
        errs["loop_vs_stage"] = _rel(y, m.transformer_blocks(x.to(dev())).double().cpu())
    report("submodules_transformer", **errs)
    assert max(errs.values()) < 2e-5, errs


def test_submodules_under_autocast_and_without_partial_transformers():
    from oracle import beat_this_oracle as O

    m, sd, h = _model()
    g = torch.Generator().manual_seed(4)
    x = torch.randn((1, 96, h["transformer_dim"]), generator=g)
    with torch.inference_mode(), torch.autocast("cuda"):
        a = m.transformer_blocks.layers[0][0](x.to(dev()))
        f = m.transformer_blocks.layers[0][1](x.to(dev()))
    ea = _rel(a, O.attention(x.double(), sd, "transformer_blocks.layers.0.0.", h["transformer_dim"] // 32))
    ef = _rel(f, O.feedforward(x.double(), sd, "transformer_blocks.layers.0.1."))
    m2, sd2, _ = _model(partial_transformers=False)
    s = torch.randn((1, 32, 32, 50), generator=g)
    with torch.inference_mode():
        assert "partial" not in m2.frontend.blocks[0]._modules
        b0 = m2.frontend.blocks[0](s.to(dev()))
    p = "frontend.blocks.0."
    ref = torch.nn.functional.gelu(O.batchnorm(torch.nn.functional.conv2d(s.double(), sd2[p + "conv2d.weight"], stride=(2, 1), padding=(0, 1)),
                                               sd2, p + "norm.", 1))
    eb = _rel(b0, ref)
    report("submodules_misc", attn_half=ea, ff_half=ef, block_without_partial=eb)
    assert ea < 2e-2 and ef < 2e-2 and eb < 2e-5
    with pytest.raises(ValueError):
        m.frontend.blocks[1](s.to(dev()))            # block 1 takes (b, 64, 16, t)
    with pytest.raises(NotImplementedError):
        m.frontend.blocks[0].partial.attnF.to_qkv(s.to(dev()))    # (parameter-only nodes below the callable leaves)
    with pytest.raises(ValueError):
        m.frontend.blocks[0].partial.attnF(s.to(dev()))           # (sequences, tokens, 32) expected


def test_partial_transformer_leaves_match_the_oracle():
    """partial.attnF / .ffF / .attnT / .ffT are ordinary Attention / FeedForward modules in the reference (beat_tracker.py:251-301),
    called on "(b t) f c" / "(b f) t c" rows: each against the oracle's operator, and the reference's own composition of the four
    (with its rearranges) against the .partial unit."""
    from oracle import beat_this_oracle as O

    m, sd, _ = _model()
    g = torch.Generator().manual_seed(9)
    errs = {}
    with torch.inference_mode():
        for i in range(3):
            c, f, t, b = 32 << i, 32 >> i, 70, 2
            p = f"frontend.blocks.{i}.partial."
            part = m.frontend.blocks[i].partial
            xf = torch.randn((b * t, f, c), generator=g)
            xt = torch.randn((b * f, t, c), generator=g)
            errs[f"attnF{i}"] = _rel(part.attnF(xf.to(dev())), O.attention(xf.double(), sd, p + "attnF.", c // 32))
            errs[f"ffF{i}"] = _rel(part.ffF(xf.to(dev())), O.feedforward(xf.double(), sd, p + "ffF."))
            errs[f"attnT{i}"] = _rel(part.attnT(xt.to(dev())), O.attention(xt.double(), sd, p + "attnT.", c // 32))
            errs[f"ffT{i}"] = _rel(part.ffT(xt.to(dev())), O.feedforward(xt.double(), sd, p + "ffT."))
            # PartialFTTransformer.forward written against this model's leaves (beat_tracker.py:290-301)
            x = torch.randn((b, c, f, t), generator=g).to(dev())
            y = x.permute(0, 3, 2, 1).reshape(b * t, f, c)
            y = y + part.attnF(y)
            y = y + part.ffF(y)
            y = y.view(b, t, f, c).permute(0, 2, 1, 3).reshape(b * f, t, c)
            y = y + part.attnT(y)
            y = y + part.ffT(y)
            y = y.view(b, f, t, c).permute(0, 3, 1, 2)
            errs[f"composed{i}"] = _rel(y, part(x).double().cpu())
    report("submodules_partial_leaves", **errs)
    assert max(errs.values()) < 2e-5, errs


def test_hooks_on_deep_submodules_fire_in_the_whole_forward():
    """A forward hook on a sub-module below the stages (here: the FeedForward of layer 1, the partial transformer of frontend
    block 2, the whole block 0) makes BeatThis.forward run through the sub-modules, so the hook fires once per forward with
    the reference's tensor shapes, and the result is that of the whole-forward call (fp32: the same kernels up to the fused norm + head)."""
    from beat_this_amd import weights as W

    m, _, h = _model()
    x = torch.from_numpy(W.synthetic_spect(150, seed=7))[None].to(dev())
    with torch.inference_mode():
        plain = m(x)
    seen = {}
    def rec(name):
        def hook(mod, inp, out):   # (returns None: a forward hook that returns something replaces the output)
            seen.setdefault(name, (tuple(inp[0].shape), tuple(out.shape)))
        return hook
    hs = [m.transformer_blocks.layers[1][1].register_forward_hook(rec("ff1")),
          m.frontend.blocks[2].partial.register_forward_hook(rec("partial2")),
          m.frontend.blocks[0].register_forward_hook(rec("block0"))]
    with torch.inference_mode():
        hooked = m(x)
    for hd in hs:
        hd.remove()
    D = h["transformer_dim"]
    assert seen == {"ff1": ((1, 150, D), (1, 150, D)), "partial2": ((1, 128, 8, 150), (1, 128, 8, 150)),
                    "block0": ((1, 32, 32, 150), (1, 64, 16, 150))}, seen
    # (stage by stage the final RMSNorm and the head are two kernels instead of one: agreement to fp32 rounding)
    assert float((hooked["beat"] - plain["beat"]).abs().max()) < 1e-5 and float((hooked["downbeat"] - plain["downbeat"]).abs().max()) < 1e-5
    with torch.inference_mode():
        again = m(x)   # hooks removed: the fast path again
        m.fp32_split_gemms = True
        default = m(x)   # the module's default precision against the chain of exact sub-modules: fp32-class agreement
    assert torch.equal(again["beat"], plain["beat"])
    assert 0 < float((default["beat"] - hooked["beat"]).abs().max()) < 3e-4
