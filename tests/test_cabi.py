"""No-GPU checks of the C-ABI library: it builds for gfx950, loads next to torch's ROCm runtime,
exports every symbol include/beat_this_amd.h declares, and its host-only entry point
(bt_postprocess_host) matches the oracle and the reference's golden outputs bit for bit."""
import ctypes as C
import json
import os
import re

import numpy as np
import torch

from conftest import GOLDEN, ROOT
from oracle import beat_this_oracle as O
from oracle.cases import POSTP_CASES


def _lib():
    from beat_this_amd import _lib as L

    L.build()
    return L


def test_library_exports_every_declared_symbol():
    L = _lib()
    header = open(os.path.join(ROOT, "include", "beat_this_amd.h")).read()
    declared = set(re.findall(r"\b(bt_[a-z_0-9]+)\s*\(", header))
    assert re.search(r"#define BT_ABI_VERSION 600\b", header)
    assert {"bt_forward", "bt_logmel", "bt_peaks", "bt_aggregate", "bt_split_chunks", "bt_engine_create"} <= declared
    handle = L.lib()
    for name in declared:
        assert hasattr(handle, name), f"{name} declared in the header but not exported"
    assert set(L.EXPORTS) == declared
    assert handle.bt_version() == L.ABI_VERSION == 600   # (BT_ABI_VERSION of include/beat_this_amd.h)


def test_argument_errors_map_to_exceptions():
    L = _lib()
    import pytest

    rc = L.lib().bt_forward(None, None, 0, None, 1, 1, None, 0, None, None)
    assert rc == L.BT_ERR_ARG
    with pytest.raises(ValueError):
        L.check(rc)
    assert L.lib().bt_workspace_bytes(None, 1, 1, 0) == 0
    # the staged entry point rejects its arguments before touching a GPU as well
    assert L.lib().bt_forward_stages(None, None, 0, 0, 2, None, 1, 1, None, 0, None, None, None) == L.BT_ERR_ARG
    assert b"null" in L.lib().bt_last_error()


def test_struct_layout_matches_header():
    L = _lib()
    sizes = (C.c_int32 * 9)()
    L.lib().bt_struct_sizes(sizes)
    assert list(sizes) == [C.sizeof(L.PairWeights), C.sizeof(L.ModelDesc), C.sizeof(L.LogmelTables),
                           C.sizeof(L.GemmArgs), C.sizeof(L.AttnArgs), L.ModelDesc.layers.offset,
                           L.ModelDesc.rope.offset, C.sizeof(L.AttnFragArgs), C.sizeof(L.Gemm3Args)]


def test_host_postprocess_bit_exact():
    from beat_this_amd.postprocessor import _host_post

    post = json.load(open(os.path.join(GOLDEN, "postp_minimal.json")))
    for name, (bs, ds) in POSTP_CASES.items():
        b = torch.full((100,), -5.0)
        d = torch.full((100,), -5.0)
        for f, v in bs:
            b[f] = v
        for f, v in ds:
            d[f] = v
        bt, dt = _host_post(O.peak_frames(b), O.peak_frames(d), 50)
        assert bt.tolist() == post[name]["beats"], name
        assert dt.tolist() == post[name]["downbeats"], name
    rng = np.random.default_rng(123)
    for _ in range(20):
        n = int(rng.integers(1, 4000))
        b = torch.from_numpy(rng.normal(-0.5, 1.5, n).astype(np.float32))
        d = torch.from_numpy(rng.normal(-1.5, 1.5, n).astype(np.float32))
        b[rng.integers(0, n, n // 10)] = 1.25  # plateaus / equal neighbours
        bt, dt = _host_post(O.peak_frames(b), O.peak_frames(d), 50)
        ob, od = O.postp_minimal(b, d)
        assert np.array_equal(bt, ob) and np.array_equal(dt, od)
    bt, dt = _host_post(np.zeros(0, np.int32), np.array([5, 6, 30], np.int32), 50)
    assert len(bt) == 0 and dt.tolist() == [5.5 / 50, 30 / 50]


def test_package_fails_loudly_without_gpu():
    import pytest

    from beat_this_amd.inference import Spect2Frames
    from beat_this_amd.model import BeatThis
    from beat_this_amd.preprocessing import LogMelSpect

    with pytest.raises(RuntimeError):
        BeatThis()(torch.zeros(1, 50, 128))
    with pytest.raises(RuntimeError):
        LogMelSpect()(torch.zeros(4000))
    with pytest.raises(RuntimeError, match="no CPU implementation"):
        Spect2Frames(checkpoint_path=None, device="cpu")  # refused in __init__, not at the first call


def test_host_peak_mask_and_cpu_logits_postprocessor():
    """Postprocessor on CPU logits (the reference accepts them, postprocessor.py:58-83): bt_peaks_host + bt_postprocess_host
    through the C ABI, bit-exact against the oracle and the reference's golden edge cases."""
    from beat_this_amd.postprocessor import Postprocessor

    post = json.load(open(os.path.join(GOLDEN, "postp_minimal.json")))
    pp = Postprocessor("minimal", fps=50)
    for name, (bs, ds) in POSTP_CASES.items():
        b = torch.full((100,), -5.0)
        d = torch.full((100,), -5.0)
        for f, v in bs:
            b[f] = v
        for f, v in ds:
            d[f] = v
        bt, dt = pp(b, d)
        assert bt.tolist() == post[name]["beats"] and dt.tolist() == post[name]["downbeats"], name
    rng = np.random.default_rng(post["random3000"]["seed"])
    rb = torch.from_numpy(rng.normal(-1.0, 1.5, 3000).astype(np.float32))
    rd = torch.from_numpy(rng.normal(-2.0, 1.5, 3000).astype(np.float32))
    bt, dt = pp(rb, rd)
    assert bt.tolist() == post["random3000"]["beats"] and dt.tolist() == post["random3000"]["downbeats"]
    bb, dd = pp(torch.stack([rb, rb]), torch.stack([rd, rd]))
    assert isinstance(bb, tuple) and bb[1].tolist() == post["random3000"]["beats"]
    # padding mask path (postprocessor.py:100-104,119-120)
    mask = torch.ones(3000, dtype=torch.bool)
    mask[2500:] = False
    mb, md = pp(rb, rd, mask)
    ob, od = O.postp_minimal(rb[:2500].clone().masked_fill(~mask[:2500], -1000.0), rd[:2500])
    assert np.array_equal(mb, ob) and np.array_equal(md, od)


def _check_padding_mask_cases(device):
    """``Postprocessor(beat, downbeat, padding_mask)`` against the UNMODIFIED reference's outputs on the same seeded inputs
    (tests/golden/postp_padding_mask.json, oracle/make_golden_padding.py): tail padding, masked stretches in front of and
    between valid frames (the indices behind them shift, postprocessor.py:113-117), a fully masked row, unbatched call."""
    from beat_this_amd.postprocessor import Postprocessor
    from oracle.cases import PADDING_MASK_CASES, padding_mask_case

    gold = json.load(open(os.path.join(GOLDEN, "postp_padding_mask.json")))
    pp = Postprocessor("minimal", fps=50)
    for name in PADDING_MASK_CASES:
        beat, down, mask = padding_mask_case(name)
        bt, dt = pp(torch.from_numpy(beat).to(device), torch.from_numpy(down).to(device), torch.from_numpy(mask).to(device))
        if beat.ndim == 1:
            assert isinstance(bt, np.ndarray) and bt.dtype == np.float64
            bt, dt = (bt,), (dt,)
        else:
            assert isinstance(bt, tuple) and len(bt) == beat.shape[0]
        for r in range(len(bt)):
            assert bt[r].tolist() == gold[name]["beats"][r], (name, r)
            assert dt[r].tolist() == gold[name]["downbeats"][r], (name, r)


def test_postprocessor_padding_mask_on_host_logits_matches_reference():
    _check_padding_mask_cases("cpu")


def test_hl32_format_round_trips_and_matches_the_packer():
    """The hi / lo plane format of BT_PREC_F32X3 (csrc/gemm3.hip): x = hi + lo to 2^-22 relative (2^-25 absolute for tiny
    values), the layout is [32 hi | 32 lo] per 32 columns, and beat_this_amd.pack's weight packer and the tests' helper
    produce the same bytes."""
    import sys

    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from gpu_util import from_hl32, to_hl32

    g = torch.Generator().manual_seed(0)
    x = torch.randn((70, 96), generator=g) * torch.logspace(-6, 3, 96)[None, :]
    h = to_hl32(x)
    assert h.shape == (70, 192) and h.dtype == torch.float16
    err = (from_hl32(h) - x.double()).abs()
    assert float((err / x.double().abs().clamp_min(2.0 ** -3)).max()) <= 2.0 ** -21   # 22 bits above 2^-3, 2^-24 absolute below
    assert torch.equal(h.view(70, 3, 2, 32)[:, :, 0].reshape(70, 96), x.to(torch.float16))   # hi plane = half(x)

    class _P:   # PackedModel._hl32 without a device
        device = torch.device("cpu")
        _keep = []
    from beat_this_amd.pack import PackedModel

    L = _lib()
    if not L.lib().bt_half_is_bf16():
        w = torch.randn((130, 64), generator=g) * 0.05
        p = _P()
        assert PackedModel._hl32(p, w) != 0
        packed = p._keep[-1]
        assert packed.shape == (256, 128)   # rows padded to a multiple of 256
        assert torch.equal(packed[:130], to_hl32(w)) and torch.all(packed[130:] == 0)


def test_submodule_tree_mirrors_the_reference_containers():
    """CPU: the nodes the reference exposes as callable sub-modules are bound to engine units, index like its ModuleList /
    Sequential, survive deepcopy, leave the state dict alone and fail loudly on CPU tensors / on parameter-only nodes."""
    import copy

    import pytest
    import torch

    from beat_this_amd.model import BeatThis

    m = BeatThis(transformer_dim=128, n_layers=3)
    keys = set(m.state_dict())
    assert len(m.frontend.blocks) == 3 and len(m.transformer_blocks.layers) == 3 and len(m.transformer_blocks.layers[0]) == 2
    assert [b._unit for b in m.frontend.blocks] == [("block", 0), ("block", 1), ("block", 2)]
    assert m.transformer_blocks.layers[2][0]._unit == ("attn", 2) and m.transformer_blocks.layers[2][1]._unit == ("ff", 2)
    assert m.frontend.stem._unit == ("stem", 0) and m.frontend.concat._unit == ("concat", 0) and bool(m.frontend.concat)
    pairs = [(a._unit, f._unit) for a, f in m.transformer_blocks.layers]
    assert pairs == [(("attn", l), ("ff", l)) for l in range(3)]
    m2 = copy.deepcopy(m)
    assert m2.frontend.linear._root() is m2 and set(m2.state_dict()) == keys
    with pytest.raises(RuntimeError, match="ROCm GPUs only"):
        m.frontend.stem(torch.zeros(1, 10, 128))
    assert m.frontend.blocks[1].partial.attnT._unit == ("fattn", 3) and m.frontend.blocks[2].partial.ffF._unit == ("fff", 4)
    with pytest.raises(NotImplementedError):
        m.frontend.blocks[0].partial.attnF.norm(torch.zeros(1))   # (parameter-only nodes below the callable leaves)


def test_hubconf_exports_the_reference_entry_points():
    """hubconf.py:10-18 of the reference: `beat_this` (= load_model) and the inference classes under the same names."""
    import importlib.util

    spec = importlib.util.spec_from_file_location("bt_hubconf", os.path.join(ROOT, "hubconf.py"))
    hub = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(hub)
    from beat_this_amd import inference as inf

    assert hub.beat_this is inf.load_model
    for name in ("BeatThis", "Spect2Frames", "Audio2Frames", "Audio2Beats", "File2Beats", "File2File"):
        assert getattr(hub, name) is getattr(inf, name)
    assert set(hub.dependencies) <= {"torch", "numpy"}


def test_reference_import_paths_resolve():
    """beat_this.model is a package in the reference (model/beat_tracker.py, model/postprocessor.py; pl_module.py imports from both):
    the same dotted paths exist here and give the same classes as the flat names."""
    from beat_this_amd.inference import BeatThis as B0
    from beat_this_amd.inference import Postprocessor as P0
    from beat_this_amd.model import BeatThis as B1
    from beat_this_amd.model.beat_tracker import BeatThis as B2
    from beat_this_amd.model.postprocessor import Postprocessor as P1

    assert B0 is B1 is B2 and P0 is P1


def test_deduplicate_peaks_matches_the_oracle_restatement():
    """the reference's public helper (postprocessor.py:176-197) on the library's host entry point: plateaus, chains that drift
    with the running mean, other widths, the empty list"""
    from beat_this_amd.model.postprocessor import deduplicate_peaks

    rng = np.random.default_rng(3)
    cases = [[], [5], [10, 11], [10, 11, 12], [10, 11, 12, 13, 14, 15], [3, 5, 7], [0, 1, 3, 4, 6, 20, 21, 40],
             sorted(set(rng.integers(0, 400, 150).tolist()))]
    for peaks in cases:
        for width in (1, 2, 3):
            want = O.deduplicate_peaks(np.asarray(peaks, dtype=np.int64), width) if peaks else np.zeros(0)
            got = deduplicate_peaks(peaks, width)
            assert got.dtype == np.float64 and np.array_equal(got, np.asarray(want, dtype=np.float64)), (peaks, width)


def test_the_captured_forward_contains_no_memset_or_memcpy_nodes():
    """A captured forward must consist of kernel launches only: memset nodes at the head of a hipGraph misbehaved on the default stream
    (tests/test_gpu_model.py::test_interleaved_graph_replays_on_the_default_stream_never_raise_the_range_flag).  bt_forward_stages may
    copy device-to-device only on its partial-stage entries / exits (never part of a captured whole forward)."""
    import re

    src = open(os.path.join(ROOT, "beat_this_amd", "csrc", "engine.hip")).read()
    body = src[src.index("int bt_forward_stages("):src.index("int bt_forward_unit(") if "int bt_forward_unit(" in src else len(src)]
    assert "hipMemsetAsync" not in body, "bt_forward_stages clears its flag with launch_clear_words, not with a memset node"
    copies = re.findall(r"hipMemcpyAsync\(([^;]*);", body)
    assert all("DeviceToDevice" in c for c in copies)
