"""Spectrogram bundle reader (reference behaviour: beat_this/dataset/mmnpz.py:12-108)."""
import numpy as np
import pytest

from beat_this_amd.bundle import SpectBundle


def _arrays():
    rng = np.random.default_rng(3)
    return {"a/track1": rng.random((300, 128)).astype(np.float16), "b": rng.random((7, 128)).astype(np.float32),
            "fortran": np.asfortranarray(rng.random((5, 4))), "scalarish": np.arange(3, dtype=np.int64)}


def test_bundle_views_match_numpy_load(tmp_path):
    arrs = _arrays()
    fn = tmp_path / "bundle.npz"
    np.savez(fn, **arrs)
    with SpectBundle(fn) as bun:
        assert sorted(bun.files) == sorted(arrs) and len(bun) == len(arrs)
        assert "b" in bun and "nope" not in bun
        for k, v in arrs.items():
            got = bun[k]
            assert got.dtype == v.dtype and got.shape == v.shape and np.array_equal(got, v)
            assert not got.flags.writeable and not got.flags.owndata  # a view into the file mapping, not a copy
        assert bun["fortran"].flags.f_contiguous
        with pytest.raises(KeyError):
            bun["nope"]
    with pytest.raises(ValueError):
        bun["b"]  # closed


def test_bundle_skips_compressed_members_and_preloads(tmp_path):
    arrs = _arrays()
    fz = tmp_path / "compressed.npz"
    np.savez_compressed(fz, **arrs)
    assert SpectBundle(fz).files == []
    fn = tmp_path / "plain.npz"
    np.savez(fn, **arrs)
    bun = SpectBundle(fn, preload=True)
    assert set(bun._views) == set(arrs)
    assert dict(bun).keys() == arrs.keys()


@pytest.mark.gpu
def test_predict_bundle_matches_piecewise(tmp_path):
    import torch

    from beat_this_amd import weights as W
    from beat_this_amd.bundle import predict_bundle
    from beat_this_amd.inference import Spect2Frames

    hp = W.HPARAMS["small0"]
    ckpt = {"state_dict": {"model." + k: v for k, v in W.random_state_dict(hp, seed=1, style="lively").items()},
            "hyper_parameters": dict(hp)}
    s2f = Spect2Frames(ckpt, "cuda")
    pieces = {f"p{i}": W.synthetic_spect(n, seed=50 + i).astype(np.float16) for i, n in enumerate((700, 1600, 90, 3100))}
    fn = tmp_path / "spects.npz"
    np.savez(fn, **pieces)
    with SpectBundle(fn) as bun:
        out = {n: (b, d) for n, b, d in predict_bundle(s2f, bun, group_frames=2500)}
        assert list(out) == list(bun.files)
        for n in bun.files:
            rb, rd = s2f(torch.from_numpy(np.array(bun[n])).to("cuda").float())
            assert out[n][0].shape == (pieces[n].shape[0],)
            assert torch.allclose(out[n][0], rb, atol=2e-4) and torch.allclose(out[n][1], rd, atol=2e-4)
