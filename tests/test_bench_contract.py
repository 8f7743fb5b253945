"""bench.py's bookkeeping on CPU: the algorithmic FLOP table must reproduce SURVEY.md 8d / Appendix B, the track arithmetic
must match split_piece, and the command line must expose the driver contract."""
import importlib.util
import os
import subprocess
import sys

from conftest import ROOT


def _bench():
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_flop_table_matches_survey():
    b = _bench()
    fl = b.flops_per_chunk(512)
    total = sum(v for k, v in fl.items() if k != "layer_tail")
    assert abs(total - 134.71e9) < 0.01e9          # 67.355 GMAC = 134.71 GFLOP per final0 chunk
    assert fl["layer_tail"] == fl["out_gemm"] + fl["ff1_gemm"] + fl["ff2_gemm"]
    small = b.flops_per_chunk(128)
    assert abs(sum(v for k, v in small.items() if k != "layer_tail") - 59.57e9) < 0.01e9
    # per layer: qkv 1.1796 + gates 0.0123 + QK^T+PV 2.3040 + out 0.3932 + ff 3.1457 GMAC
    assert abs(fl["attn_flash"] - 2 * (6 * 2.3040e9 + 3 * 4.608e9)) < 1e7


def test_track_arithmetic_matches_split_piece():
    from beat_this_amd.inference import chunk_starts
    from beat_this_amd.parallel import track_frames

    assert track_frames(int(300.0 * 44100), 44100) == 15001
    assert len(chunk_starts(15001, 1500, 6)) == 11  # a 5-minute track is 11 chunks
    assert track_frames(int(300.0 * 22050), 22050) == 15001


def test_command_line_contract():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--help"], capture_output=True, text=True).stdout
    for flag in ("--gpus", "--steps", "--warmup", "--prec", "--workload", "--tracks"):
        assert flag in out
