"""Generate tests/golden/final0_piece_e2e.npz by running the UNMODIFIED reference on CPU (round 6, VERDICT r5 item 4:
the reference's own outputs for final0 at piece / end-to-end level, and an "outlier"-style chunk -- so far final0 had two
single-chunk goldens and everything larger was checked against the oracle only, one hop further from the reference).

TEST INFRASTRUCTURE ONLY.  Run in the build container (needs /root/reference):

    PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden_final0.py

Same import arrangement as oracle/make_golden.py (the reference's package from /root/reference, the three third-party
stand-ins of oracle/shims first on sys.path); a file of its own so that the existing fixtures are not regenerated.  Inputs are
regenerated from seeds (beat_this_amd.weights, numpy PCG64): the fixture holds only the reference's OUTPUTS.
"""
import os
import sys

sys.dont_write_bytecode = True
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("BEAT_THIS_REFERENCE", "/root/reference")
sys.path[:0] = [os.path.join(ROOT, "oracle", "shims"), REF, ROOT]

import numpy as np  # noqa: E402
import torch  # noqa: E402

from beat_this.inference import Audio2Beats, split_predict_aggregate  # noqa: E402
from beat_this.model.beat_tracker import BeatThis  # noqa: E402

from beat_this_amd import weights as W  # noqa: E402
from oracle.cases import FINAL0_CASES  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
MODEL_KEYS = ("spect_dim", "transformer_dim", "ff_mult", "n_layers", "head_dim", "stem_dim")


def reference_model(hp_name, seed, style):
    hp = W.resolve_hparams(hp_name)
    m = BeatThis(**{k: hp[k] for k in MODEL_KEYS}).eval()
    m.load_state_dict(W.random_state_dict(hp, seed=seed, style=style))
    return m


def main():
    torch.manual_seed(0)
    c = FINAL0_CASES
    arrs = {}
    # 1. split_predict_aggregate over a 3100-frame piece (inference.py:188-230): three chunks, the last one moved back
    m = reference_model("final0", c["piece"]["weight_seed"], c["piece"]["style"])
    piece = torch.from_numpy(W.synthetic_spect(c["piece"]["frames"], seed=c["piece"]["input_seed"]))
    with torch.inference_mode():
        r = split_predict_aggregate(piece, 1500, 6, "keep_first", m)
    arrs["piece_beat"], arrs["piece_downbeat"] = r["beat"].numpy(), r["downbeat"].numpy()
    print("piece", r["beat"].shape, float(r["beat"].std()))
    # 2. Audio2Beats end to end from a 40 s 22.05 kHz waveform (inference.py:279-303): 2001 frames = two chunks
    a2b = Audio2Beats(checkpoint_path=None, device="cpu")
    a2b.model = reference_model("final0", c["e2e"]["weight_seed"], c["e2e"]["style"])
    sig = W.synthetic_audio(c["e2e"]["seconds"], seed=c["e2e"]["audio_seed"])
    beats, downbeats = a2b(sig, 22050)
    with torch.inference_mode():
        bl, dl = a2b.spect2frames(a2b.signal2spect(sig, 22050))
    arrs.update(e2e_beats=beats, e2e_downbeats=downbeats, e2e_beat_logits=bl.numpy(), e2e_downbeat_logits=dl.numpy())
    print("e2e beats", len(beats), "downbeats", len(downbeats), "frames", bl.shape[0])
    # 3. one chunk on the trained-like "outlier" weights (residual outlier channels, heavy-tailed matrices; beat_tracker.py:188-192)
    m = reference_model("final0", c["outlier"]["weight_seed"], "outlier")
    x = torch.from_numpy(W.synthetic_spect(c["outlier"]["frames"], seed=c["outlier"]["input_seed"]))[None]
    with torch.inference_mode():
        r = m(x)
    arrs["outlier_beat"], arrs["outlier_downbeat"] = r["beat"][0].numpy(), r["downbeat"][0].numpy()
    print("outlier chunk", float(r["beat"].std()), float((r["beat"] > 0).float().mean()))
    np.savez_compressed(os.path.join(OUT, "final0_piece_e2e.npz"), **arrs)


if __name__ == "__main__":
    main()
