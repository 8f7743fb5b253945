"""Spectrogram bundles: zero-copy access to the arrays of an *uncompressed* ``.npz`` archive and batched
inference over them (the reference keeps dataset-scale spectrograms as float16 ``.npz``
bundles and memory-maps them, beat_this/dataset/mmnpz.py:12-108, README.md:92-101).

``SpectBundle`` maps the archive file once; every stored (ZIP_STORED) ``.npy`` member becomes a read-only
ndarray view into that mapping -- no decompression, no copy, float16 stays float16 until it is on the GPU.
``predict_bundle`` pushes the pieces of a bundle through ``Spect2Frames.spect2frames_many`` in groups, so
the chunks of many short pieces share forward launches (and ranks, when torch.distributed is initialised).
"""
from __future__ import annotations

import io
import struct
import warnings
import zipfile
from collections.abc import Mapping

import numpy as np

_LOCAL_HEADER = struct.Struct("<4s5H3I2H")  # signature .. name length, extra length (30 bytes)


class SpectBundle(Mapping):
    """Read-only mapping ``name -> ndarray`` over the stored ``.npy`` members of an ``.npz`` file.

    Compressed members are not listed (they cannot be mapped); ``files`` holds the usable keys in archive
    order.  Arrays are views into one ``np.memmap`` and stay valid until ``close()``."""

    def __init__(self, path, preload: bool = False):
        self.path = str(path)
        self._members: dict[str, tuple[int, int]] = {}
        with zipfile.ZipFile(self.path) as zf:
            for info in zf.infolist():
                if info.filename.endswith(".npy") and info.compress_type == zipfile.ZIP_STORED:
                    self._members[info.filename[:-4]] = (info.header_offset, info.file_size)
        self.files = list(self._members)
        self._map = np.memmap(self.path, dtype=np.uint8, mode="r")
        self._views: dict[str, np.ndarray] = {}
        if preload:
            for name in self.files:
                self[name]

    def _view(self, name: str) -> np.ndarray:
        header_offset, size = self._members[name]
        # the central directory does not know how long the LOCAL header's name / extra fields are
        fields = _LOCAL_HEADER.unpack(bytes(self._map[header_offset: header_offset + _LOCAL_HEADER.size]))
        if fields[0] != b"PK\x03\x04":
            raise ValueError(f"{self.path}: corrupt local header for member '{name}'")
        start = header_offset + _LOCAL_HEADER.size + fields[-2] + fields[-1]
        head = io.BytesIO(bytes(self._map[start: start + min(size, 4096)]))
        major, minor = np.lib.format.read_magic(head)
        if (major, minor) == (1, 0):
            shape, fortran, dtype = np.lib.format.read_array_header_1_0(head)
        elif (major, minor) in ((2, 0), (3, 0)):
            shape, fortran, dtype = np.lib.format.read_array_header_2_0(head)
        else:
            raise ValueError(f"{self.path}: unsupported .npy version {major}.{minor} in member '{name}'")
        if dtype.hasobject:
            raise ValueError(f"{self.path}: member '{name}' holds Python objects")
        data = self._map[start + head.tell(): start + size]
        return data.view(dtype).reshape(shape, order="F" if fortran else "C")

    def __getitem__(self, name: str) -> np.ndarray:
        if self._map is None:
            raise ValueError("bundle is closed")
        if name not in self._views:
            if name not in self._members:
                raise KeyError(name)
            self._views[name] = self._view(name)
        return self._views[name]

    def __iter__(self):
        return iter(self.files)

    def __len__(self):
        return len(self.files)

    def __contains__(self, name) -> bool:
        return name in self._members

    def close(self) -> None:
        self._views = {}
        self._map = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


def predict_bundle(spect2frames, bundle: Mapping, names=None, group_frames: int = 64 * 1488):
    """Framewise logits for the (frames, 128) spectrograms of ``bundle`` (any mapping name -> array, float16 or
    float32).  Pieces are uploaded as stored, widened to fp32 on the device, and sent through
    ``spect2frames.spect2frames_many`` in groups of about ``group_frames`` frames.  Yields
    ``(name, beat_logits, downbeat_logits)`` in the order of ``names`` (default: the bundle's)."""
    import torch

    names = list(bundle if names is None else names)
    i = 0
    while i < len(names):
        group, frames = [], 0
        while i < len(names) and (not group or frames + bundle[names[i]].shape[0] <= group_frames):
            group.append(names[i])
            frames += bundle[names[i]].shape[0]
            i += 1
        spects = []
        for n in group:
            a = np.ascontiguousarray(bundle[n])
            if a.ndim != 2 or a.shape[1] != 128:
                raise ValueError(f"'{n}': expected a (frames, 128) spectrogram, got {a.shape}")
            with warnings.catch_warnings():  # read-only mapping: the tensor is only a source of the upload
                warnings.simplefilter("ignore", UserWarning)
                host = torch.from_numpy(a)
            spects.append(host.to(spect2frames.device).to(torch.float32))
        for n, (b, d) in zip(group, spect2frames.spect2frames_many(spects)):
            yield n, b, d
