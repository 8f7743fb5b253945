"""Real-checkpoint leg (SURVEY.md 8c iv; reference: inference.py:13,38-48): when a trained checkpoint is available --
``$BEAT_THIS_CKPT`` (a path) or ``beat_this-final0.ckpt`` in torch.hub's checkpoint cache, where the reference's
``load_checkpoint("final0")`` leaves it -- the forward, end-to-end and command-line parity checks also run with TRAINED
weights: trained score ranges drive the attention's overflow fallback, the fp16 headroom of the probabilities and the
operand range of the hi + lo split (BT_PREC_F32X3), none of which seeded random weights exercise.  There is no network on
the build or GPU boxes, so without such a file every test here SKIPS and says so."""
import os

import numpy as np
import pytest
import torch

from conftest import have_reference

LOGIT_TOL_F32 = 1e-3


def real_checkpoint():
    """Path of a trained checkpoint, or None."""
    p = os.environ.get("BEAT_THIS_CKPT")
    if p and os.path.isfile(p):
        return p
    for name in ("final0", "small0"):
        q = os.path.join(torch.hub.get_dir(), "checkpoints", f"beat_this-{name}.ckpt")
        if os.path.isfile(q):
            return q
    return None


needs_ckpt = pytest.mark.skipif(real_checkpoint() is None, reason="no trained checkpoint here: set $BEAT_THIS_CKPT or put "
                                "beat_this-final0.ckpt into torch.hub's checkpoint cache (no network: cannot download)")


def _state():
    from beat_this_amd.inference import load_checkpoint
    from beat_this_amd.utils import replace_state_dict_key

    ckpt = load_checkpoint(real_checkpoint(), "cpu")
    sd = replace_state_dict_key(dict(ckpt["state_dict"]), "model.", "")
    return ckpt, {k.replace("_orig_mod.", ""): v for k, v in sd.items()}


@needs_ckpt
def test_oracle_matches_live_reference_on_the_trained_checkpoint():
    """CPU: the oracle restatement against the unmodified reference with the TRAINED weights (one 1500-frame chunk)."""
    if not have_reference():
        pytest.skip("needs /root/reference (build container only)")
    import sys

    from beat_this_amd import weights as W
    from oracle import beat_this_oracle as O

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path[:0] = [os.path.join(root, "oracle", "shims"), "/root/reference"]
    from beat_this.inference import load_model as ref_load_model

    _, sd = _state()
    ref = ref_load_model(real_checkpoint(), "cpu")
    x = torch.from_numpy(W.synthetic_spect(1500, seed=3))[None]
    with torch.inference_mode():
        r = ref(x)
        ob, od = O.model_forward(sd, x)
    assert float((r["beat"] - ob).abs().max()) < 5e-5 and float((r["downbeat"].float() - od).abs().max()) < 5e-5


@needs_ckpt
@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["fp32", "f32x3", "half"])
def test_forward_on_the_trained_checkpoint(mode):
    """GPU forward with trained weights against the oracle: the 1e-3 gate for the fp32 and the hi + lo split paths (plus the
    split path's range guard staying silent), the reference's own autocast error scale for the half path."""
    from beat_this_amd import weights as W
    from beat_this_amd.inference import load_model
    from gpu_util import dev, report
    from oracle import beat_this_oracle as O

    _, sd = _state()
    m = load_model(real_checkpoint(), dev())
    m.fp32_split_gemms = mode == "f32x3"
    x = torch.from_numpy(np.stack([W.synthetic_spect(1500, seed=3 + i) for i in range(2)]))
    with torch.inference_mode(), torch.autocast("cuda", enabled=mode == "half"):
        r = m(x.to(dev()))
    with torch.inference_mode():
        ob, od = O.model_forward(sd, x)
    err = max(float((r["beat"].cpu() - ob).abs().max()), float((r["downbeat"].float().cpu() - od).abs().max()))
    report("trained_checkpoint_forward", mode=mode, max_abs_logit=err, checkpoint=os.path.basename(real_checkpoint()))
    assert err < (2e-2 if mode == "half" else LOGIT_TOL_F32)
    if mode == "f32x3":
        assert m.engine().last_fallbacks == 0, "trained activations left the fp16 range of the hi + lo split"


@needs_ckpt
@pytest.mark.gpu
def test_audio2beats_on_the_trained_checkpoint():
    """End to end (40 s of the seeded click track, 22.05 kHz) with trained weights: identical beat / downbeat times."""
    from beat_this_amd import weights as W
    from beat_this_amd.inference import Audio2Beats
    from gpu_util import dev
    from oracle import beat_this_oracle as O

    _, sd = _state()
    a2b = Audio2Beats(real_checkpoint(), dev(), float16=False)
    sig = W.synthetic_audio(40.0, seed=11)
    beats, downbeats = a2b(sig, 22050)
    with torch.inference_mode():
        ob, od = O.audio2beats(sd, sig)
    assert np.array_equal(beats, ob) and np.array_equal(downbeats, od)
