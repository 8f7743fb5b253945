#!/usr/bin/env python3
"""Command line front end with the option surface of the reference's ``beat_this`` console script
(beat_this/cli.py:22-191): same positional inputs and flags, same output-path rules, same ``.beats`` TSV and
``--activations`` ``.npy`` outputs -- running on the MI355X kernels.  SURVEY.md section 8(f3).

Differences: ``--gpu -1`` (CPU) is refused, there is no CPU path in this package; with a ROCm device present
``--gpu N`` selects ``cuda:N`` exactly like the reference.
"""
from __future__ import annotations

import argparse
import sys
from pathlib import Path

import numpy as np
import torch

OPTIONS = (
    # flags, kwargs  (help texts paraphrase the reference's --help)
    (("--model",), dict(type=str, default="final0", help="checkpoint name, path or URL (default: %(default)s)")),
    (("--output", "-o"), dict(type=str, default=None,
                              help="output file (single input) or output directory (several inputs); default: next "
                                   "to each input, see --suffix / --append")),
    (("--suffix", "-s"), dict(type=str, default=".beats", help="output suffix (default: %(default)s)")),
    (("--append",), dict(action="store_true", help="append the suffix instead of replacing the input's suffix")),
    (("--skip-existing",), dict(action="store_true", help="do not overwrite existing outputs")),
    (("--touch-first",), dict(action="store_true",
                              help="create the (empty) output before processing: with --skip-existing several "
                                   "processes can share one file set")),
    (("--dbn",), dict(default=False, action=argparse.BooleanOptionalAction, help="madmom DBN post-processing")),
    (("--gpu",), dict(type=int, default=0, help="index of the GPU to use (default: %(default)s)")),
    (("--float16",), dict(action="store_true", help="half-precision path: fp16 MFMA operands, fp32 accumulation (float16=True of the Python API)")),
    (("--activations",), dict(action="store_true", help="also save the raw logits as .npy")),
)


def get_parser() -> argparse.ArgumentParser:
    parser = argparse.ArgumentParser(description="Detect beats and downbeats in audio files (Beat This!, MI355X kernels).")
    parser.add_argument("inputs", type=str, nargs="+", help="audio files and/or directories of audio files")
    for flags, kwargs in OPTIONS:
        parser.add_argument(*flags, **kwargs)
    return parser


def derive_output_path(input_path: Path, suffix: str, append: bool, output: Path | None = None,
                       parent: Path | None = None) -> Path:
    """Where the result of ``input_path`` goes (cli.py:94-112): next to the input, or under ``output`` keeping
    the path relative to the directory ``parent`` given on the command line; the suffix replaces the input's
    suffix unless ``append``."""
    if output is None:
        target = input_path
    else:
        target = output / (input_path.relative_to(parent) if parent is not None else input_path.name)
    return target.parent / (target.name + suffix) if append else target.with_suffix(suffix)


def collect_tasks(inputs, output, suffix, append, skip_existing):
    """[(audio file, output file)] for files and (recursively) directories, cli.py:163-176."""
    tasks = []
    for item in inputs:
        if item.is_dir():
            for fn in sorted(item.rglob("*")):
                if fn.is_dir() or fn.name.endswith(suffix):
                    continue
                dest = derive_output_path(fn, suffix, append, output, parent=item)
                if skip_existing and dest.exists():
                    continue
                tasks.append((fn, dest))
        else:
            tasks.append((item, derive_output_path(item, suffix, append, output)))
    return tasks


def run(inputs, model, output, suffix, append, skip_existing, touch_first, dbn, gpu, float16, activations):
    from .inference import File2File, load_audio
    from .utils import save_beat_tsv

    if gpu < 0 or not torch.cuda.is_available():
        raise SystemExit("beat_this_amd needs a ROCm GPU (there is no CPU path in this package)")
    file2file = File2File(model, torch.device(f"cuda:{gpu}"), float16, dbn)

    def process(audiofile: Path, outfile: Path):
        if not activations:
            return file2file(audiofile, outfile)
        signal, sr = load_audio(audiofile)
        beat_logits, downbeat_logits = file2file.spect2frames(file2file.signal2spect(signal, sr))
        np.save(outfile.with_suffix(".npy"), np.vstack([beat_logits.cpu().numpy(), downbeat_logits.cpu().numpy()]))
        beats, downbeats = file2file.frames2beats(beat_logits, downbeat_logits)
        save_beat_tsv(beats, downbeats, outfile)

    inputs = [Path(p) for p in inputs]
    output = Path(output) if output is not None else None
    if len(inputs) == 1 and not inputs[0].is_dir():  # single file: --output may name the file itself
        dest = output if output is not None and not output.is_dir() else derive_output_path(inputs[0], suffix, append, output)
        process(inputs[0], dest)
        return
    tasks = collect_tasks(inputs, output, suffix, append, skip_existing)
    try:
        import tqdm
        tasks = tqdm.tqdm(tasks)
    except ImportError:
        pass
    for audiofile, dest in tasks:
        if touch_first:
            try:
                dest.touch(exist_ok=not skip_existing)
            except FileExistsError:
                continue
        elif skip_existing and dest.exists():
            continue
        try:
            process(audiofile, dest)
        except Exception:  # keep going, like the reference (cli.py:185-191)
            print(f'Could not process "{audiofile}". Rerun with this file alone for details.', file=sys.stderr)


def main():
    run(**vars(get_parser().parse_args()))


if __name__ == "__main__":
    sys.exit(main())
