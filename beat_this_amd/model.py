"""``BeatThis`` -- drop-in for beat_this.model.beat_tracker.BeatThis (beat_tracker.py:18-203).

Same constructor signature, same ``state_dict`` keys (so reference checkpoints load with
``load_state_dict``), same ``forward(x: (B,T,128)) -> {"beat": (B,T), "downbeat": (B,T)}``;
the arithmetic runs in the hand-written HIP kernels of libbeat_this_amd.so.  Precision
follows the caller exactly like the reference: under ``torch.autocast`` (what
``Spect2Frames(float16=True)`` enters, inference.py:246) the half-precision (fp16 MFMA operand) path runs, otherwise
the exact-fp32 MFMA path.  There is no CPU implementation here.
"""
from __future__ import annotations

import weakref

import torch
from torch import nn

from . import _lib
from .pack import Engine, PackedModel
from .weights import random_state_dict, resolve_hparams, state_dict_shapes

_BUFFER_LEAVES = ("running_mean", "running_var", "num_batches_tracked")


class _Node(nn.Module):
    """Plain container; exists only so parameters carry the reference's dotted names."""


class _Stage(_Node):
    """``BeatThis.frontend`` / ``.transformer_blocks`` / ``.task_heads`` (beat_tracker.py:54-106): parameter containers
    with the reference's names that can also be CALLED like the reference's sub-modules -- the stage runs in the owning
    model's engine (bt_forward_stages), so forward hooks and partial forwards written against the reference keep working:
    frontend (B,T,128) -> (B,T,D); transformer_blocks (B,T,D) -> (B,T,D) incl. its final RMSNorm; task_heads (B,T,D) ->
    {"beat", "downbeat"}."""

    def __init__(self, root: "BeatThis", stage: int):
        super().__init__()
        object.__setattr__(self, "_root", weakref.ref(root))   # (not a registered sub-module: no cycle)
        self._stage = stage

    def __getstate__(self):  # (copy.deepcopy / pickle: the owner re-binds itself, BeatThis.__setstate__)
        state = self.__dict__.copy()
        state.pop("_root", None)
        return state

    def forward(self, x: torch.Tensor):
        out = self._root()._run(x, self._stage, self._stage)
        return {"beat": out[0], "downbeat": out[1]} if self._stage == 2 else out


def _attach(root: nn.Module, key: str, value: torch.Tensor) -> None:
    *path, leaf = key.split(".")
    node = root
    for part in path:
        if part not in node._modules:
            node.add_module(part, _Node())
        node = node._modules[part]
    if leaf in _BUFFER_LEAVES:
        node.register_buffer(leaf, value)
    else:
        node.register_parameter(leaf, nn.Parameter(value, requires_grad=False))


class BeatThis(nn.Module):
    def __init__(self, spect_dim: int = 128, transformer_dim: int = 512, ff_mult: int = 4, n_layers: int = 6,
                 head_dim: int = 32, stem_dim: int = 32, dropout: dict = {"frontend": 0.1, "transformer": 0.2},
                 sum_head: bool = True, partial_transformers: bool = True):
        super().__init__()
        self.hparams = resolve_hparams(dict(
            spect_dim=spect_dim, transformer_dim=transformer_dim, ff_mult=ff_mult, n_layers=n_layers,
            head_dim=head_dim, stem_dim=stem_dim, sum_head=sum_head, partial_transformers=partial_transformers))
        # reference init statistics (beat_tracker.py:170-186); inference-only, so plain tensors
        init = random_state_dict(self.hparams, seed=0, style="init0")
        for i, name in enumerate(("frontend", "transformer_blocks", "task_heads")):
            self.add_module(name, _Stage(self, i))
        for key in state_dict_shapes(self.hparams):
            _attach(self, key, init[key])
        self._engine = None
        # extension: outside autocast, run every product of the fp32 path on three half MFMAs (operands split into hi + lo
        # halves, BT_PREC_F32X3) instead of fp32 MFMAs: fp32-class results (1e-5 at the logits, identical beats) at 16/3 of
        # the matrix rate.  Operands beyond the fp16 range of a hi part are detected and the batch is repeated on the
        # exact path (Engine.forward_stages).  Off by default: the default fp32 path is the exact one.
        self.fp32_split_gemms = False
        self.eval()

    def __getstate__(self):
        state = self.__dict__.copy()
        state["_engine"] = None   # (a handle of the HIP library: rebuilt on first use)
        return state

    def __setstate__(self, state):
        super().__setstate__(state)
        for name in ("frontend", "transformer_blocks", "task_heads"):
            object.__setattr__(self._modules[name], "_root", weakref.ref(self))

    # -- state dict plumbing (beat_tracker.py:194-203: strip torch.compile's "_orig_mod.") ----
    def load_state_dict(self, state_dict, strict: bool = True, assign: bool = False):
        state_dict = {k.replace("_orig_mod.", ""): v for k, v in state_dict.items()}
        self._engine = None
        return super().load_state_dict(state_dict, strict=strict, assign=assign)

    def _apply(self, fn, *args, **kwargs):
        self._engine = None
        return super()._apply(fn, *args, **kwargs)

    @property
    def device(self) -> torch.device:
        return self.task_heads.beat_downbeat_lin.weight.device

    def engine(self) -> Engine:
        if self._engine is None:
            dev = self.device
            if dev.type != "cuda":
                raise RuntimeError(
                    f"beat_this_amd.BeatThis has no CPU implementation (parameters are on '{dev}'); "
                    "move the model to a ROCm GPU: model.to('cuda')")
            _lib.lib()  # fail loudly if the HIP library is missing
            self._engine = Engine(PackedModel(self.state_dict(), self.hparams, dev))
        return self._engine

    def forward(self, x: torch.Tensor) -> dict:
        if x.dim() != 3:
            raise ValueError(f"expected (batch, time, {self.hparams['spect_dim']}) input, got {tuple(x.shape)}")
        _lib.require_gpu(x, "model input")
        if x.shape[0] == 0 or x.shape[1] == 0:
            empty = torch.empty((x.shape[0], x.shape[1]), dtype=torch.float32, device=x.device)
            return {"beat": empty, "downbeat": empty.clone()}
        stages = (self.frontend, self.transformer_blocks, self.task_heads)
        if any(st._forward_hooks or st._forward_pre_hooks for st in stages):
            # somebody hooked a sub-module: run the stages through the modules so the hooks fire (beat_tracker.py:188-192)
            return self.task_heads(self.transformer_blocks(self.frontend(x)))
        beat, down = self._run(x, 0, 2)
        return {"beat": beat, "downbeat": down}

    def _run(self, x: torch.Tensor, first: int, last: int):
        """Stages first..last in the engine; precision follows autocast like the whole forward."""
        if x.dim() != 3:
            raise ValueError(f"expected a (batch, time, features) input, got {tuple(x.shape)}")
        _lib.require_gpu(x, "stage input")
        if x.shape[0] == 0 or x.shape[1] == 0:
            D = self.hparams["transformer_dim"]
            empty = torch.empty((x.shape[0], x.shape[1]), dtype=torch.float32, device=x.device)
            return (empty, empty.clone()) if last == 2 else torch.empty((x.shape[0], x.shape[1], D), dtype=torch.float32, device=x.device)
        half = torch.is_autocast_enabled("cuda") if hasattr(torch, "is_autocast_enabled") else False
        if half:
            prec = _lib.PREC_HALF
        else:
            prec = _lib.PREC_F32X3 if self.fp32_split_gemms else _lib.PREC_F32
        return self.engine().forward_stages(x, prec, first, last)
