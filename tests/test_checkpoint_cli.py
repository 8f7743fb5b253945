"""Checkpoint FILES and the command line.

CPU part: a Lightning-layout ``.ckpt`` written to disk loads by path, by short name (torch.hub cache) and with
torch.compile's ``_orig_mod.`` prefix (reference: inference.py:16-87, beat_tracker.py:194-203,
launch_scripts/clean_checkpoints.py:18-28); the oracle and the host TSV writer reproduce the reference CLI's golden
outputs (tests/golden/cli_small0.*, written by oracle/make_golden.py from the unmodified reference's cli.run).
GPU part: ``beat_this_amd.cli.run`` end to end on the same WAV + checkpoint file against those goldens."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN
from beat_this_amd import weights as W
from oracle.cases import CLI_CASE, lightning_checkpoint, pcm16_wav


def _golden():
    text = open(os.path.join(GOLDEN, "cli_small0.beats")).read()
    act = np.load(os.path.join(GOLDEN, "cli_small0_activations.npz"))["activations"]
    return text, act


def _write_case(tmp_path, compiled=False):
    ck = tmp_path / ("compiled.ckpt" if compiled else "plain.ckpt")
    torch.save(lightning_checkpoint(CLI_CASE["hparams"], CLI_CASE["weight_seed"], CLI_CASE["style"], compiled), ck)
    (tmp_path / "audio").mkdir(exist_ok=True)
    wav = tmp_path / "audio" / "clicks.wav"
    pcm = pcm16_wav(wav, CLI_CASE["seconds"], CLI_CASE["audio_seed"], CLI_CASE["sr"])
    return ck, wav, pcm


def test_checkpoint_file_loads_by_path_short_name_and_compiled_prefix(tmp_path, monkeypatch):
    from beat_this_amd.inference import load_checkpoint, load_model

    ck, _, _ = _write_case(tmp_path)
    ckc, _, _ = _write_case(tmp_path, compiled=True)
    src = W.random_state_dict(CLI_CASE["hparams"], seed=CLI_CASE["weight_seed"], style=CLI_CASE["style"])
    d = load_checkpoint(str(ck))
    assert set(d) >= {"state_dict", "hyper_parameters"} and all(k.startswith("model.") for k in d["state_dict"])
    for path in (ck, ckc):
        m = load_model(str(path), "cpu")
        assert m.hparams["transformer_dim"] == 128 and not m.training
        got = m.state_dict()
        assert list(got) == list(src)
        assert all(torch.equal(got[k], src[k]) for k in src)
    # short name: resolved through torch.hub's checkpoint cache as beat_this-<name>.ckpt (inference.py:36-47)
    hub = tmp_path / "torch_home"
    (hub / "hub" / "checkpoints").mkdir(parents=True)
    torch.save(torch.load(ck, weights_only=True), hub / "hub" / "checkpoints" / "beat_this-unit_test_model.ckpt")
    monkeypatch.setenv("TORCH_HOME", str(hub))
    m = load_model("unit_test_model", "cpu")
    assert torch.equal(m.state_dict()["frontend.linear.weight"], src["frontend.linear.weight"])
    # anything unloadable: the reference's ValueError (inference.py:49-53)
    with pytest.raises(ValueError, match="Could not load the checkpoint"):
        load_checkpoint(str(tmp_path / "missing" / "nothing.ckpt"))


def test_load_audio_decodes_pcm_like_the_reference(tmp_path):
    from beat_this_amd.preprocessing import load_audio

    _, wav, pcm = _write_case(tmp_path)
    sig, sr = load_audio(wav)
    assert sr == 22050 and sig.dtype == np.float64 and sig.shape == pcm.shape
    assert np.array_equal(sig, pcm.astype(np.float64) / 32768.0)
    with pytest.raises(RuntimeError, match="Could not load audio"):
        load_audio(tmp_path / "nope.wav")


def test_oracle_and_tsv_writer_reproduce_the_reference_cli_golden(tmp_path):
    from beat_this_amd.utils import save_beat_tsv
    from oracle import beat_this_oracle as O

    text, act = _golden()
    _, wav, pcm = _write_case(tmp_path)
    sd = W.random_state_dict(CLI_CASE["hparams"], seed=CLI_CASE["weight_seed"], style=CLI_CASE["style"])
    sig = (pcm.astype(np.float64) / 32768.0).astype(np.float32)
    with torch.inference_mode():
        bl, dl = O.spect2frames(sd, O.logmel(torch.from_numpy(sig)))
    assert act.shape == (2, bl.shape[0])
    assert np.abs(bl.numpy() - act[0]).max() < 1e-4 and np.abs(dl.numpy() - act[1]).max() < 1e-4
    beats, downbeats = O.postp_minimal(torch.from_numpy(act[0]), torch.from_numpy(act[1]))
    out = tmp_path / "o.beats"
    save_beat_tsv(beats, downbeats, out)
    assert out.read_text() == text
    with pytest.raises(ValueError, match="Not all downbeats are beats"):
        save_beat_tsv(np.array([0.5, 1.0]), np.array([0.75]), tmp_path / "bad.beats")


@pytest.mark.gpu
@pytest.mark.parametrize("compiled", [False, True])
def test_cli_end_to_end_matches_reference_cli_golden(tmp_path, compiled):
    from beat_this_amd import cli
    from gpu_util import report

    text, act = _golden()
    ck, wav, _ = _write_case(tmp_path, compiled)
    out = tmp_path / "out" / "clicks.beats"
    out.parent.mkdir()
    cli.run(inputs=[str(wav)], model=str(ck), output=str(out), suffix=".beats", append=False, skip_existing=False,
            touch_first=False, dbn=False, gpu=0, float16=False, activations=True)
    got = np.load(out.with_suffix(".npy"))
    err = float(np.abs(got - act).max())
    report("cli_e2e", compiled=compiled, err_activations=err, lines=len(text.splitlines()))
    assert got.shape == act.shape and err < 1e-3
    assert out.read_text() == text
    # directory mode: outputs next to --output keeping relative paths, --skip-existing honoured (cli.py:163-191)
    out2 = tmp_path / "out2"
    out2.mkdir()  # (--touch-first touches the output before anything creates its directory: the reference does the same)
    cli.run(inputs=[str(wav.parent)], model=str(ck), output=str(out2), suffix=".beats", append=False, skip_existing=True,
            touch_first=True, dbn=False, gpu=0, float16=False, activations=False)
    assert (out2 / "clicks.beats").read_text() == text


@pytest.mark.gpu
def test_file2beats_from_checkpoint_path_matches_oracle(tmp_path):
    from beat_this_amd.inference import File2Beats
    from oracle import beat_this_oracle as O

    ck, wav, pcm = _write_case(tmp_path, compiled=True)
    f2b = File2Beats(str(ck), "cuda:0", float16=False, dbn=False)
    beats, downbeats = f2b(str(wav))
    _, act = _golden()
    ob, od = O.postp_minimal(torch.from_numpy(act[0]), torch.from_numpy(act[1]))
    assert np.array_equal(beats, ob) and np.array_equal(downbeats, od)
