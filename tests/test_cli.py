"""CLI option surface and output-path rules (beat_this/cli.py:22-112), no GPU needed."""
from pathlib import Path

from beat_this_amd.cli import collect_tasks, derive_output_path, get_parser


def test_parser_matches_reference_surface():
    a = get_parser().parse_args(["a.wav", "dir", "-o", "out", "-s", ".b", "--append", "--skip-existing", "--touch-first",
                                 "--no-dbn", "--gpu", "1", "--float16", "--activations", "--model", "small0"])
    assert vars(a) == dict(inputs=["a.wav", "dir"], model="small0", output="out", suffix=".b", append=True,
                           skip_existing=True, touch_first=True, dbn=False, gpu=1, float16=True, activations=True)
    d = vars(get_parser().parse_args(["x.mp3"]))
    assert d["model"] == "final0" and d["suffix"] == ".beats" and d["gpu"] == 0 and d["dbn"] is False


def test_derive_output_path_rules():
    p = Path("/data/set/a/song.flac")
    assert derive_output_path(p, ".beats", False) == Path("/data/set/a/song.beats")
    assert derive_output_path(p, ".beats", True) == Path("/data/set/a/song.flac.beats")
    assert derive_output_path(p, ".beats", False, Path("/out")) == Path("/out/song.beats")
    assert derive_output_path(p, ".beats", False, Path("/out"), parent=Path("/data/set")) == Path("/out/a/song.beats")


def test_collect_tasks_walks_directories(tmp_path):
    (tmp_path / "in" / "sub").mkdir(parents=True)
    for n in ("in/x.wav", "in/sub/y.wav", "in/x.beats"):
        (tmp_path / n).write_bytes(b"")
    tasks = collect_tasks([tmp_path / "in"], tmp_path / "out", ".beats", False, skip_existing=False)
    assert sorted(t[1].relative_to(tmp_path).as_posix() for t in tasks) == ["out/sub/y.beats", "out/x.beats"]
    (tmp_path / "out").mkdir()
    (tmp_path / "out" / "x.beats").write_bytes(b"")
    tasks = collect_tasks([tmp_path / "in"], tmp_path / "out", ".beats", False, skip_existing=True)
    assert [t[0].name for t in tasks] == ["y.wav"]
