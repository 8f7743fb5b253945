"""``BeatThis`` -- drop-in for beat_this.model.beat_tracker.BeatThis (beat_tracker.py:18-203).

Same constructor signature, same ``state_dict`` keys (so reference checkpoints load with
``load_state_dict``), same ``forward(x: (B,T,128)) -> {"beat": (B,T), "downbeat": (B,T)}``;
the arithmetic runs in the hand-written HIP kernels of libbeat_this_amd.so.  (A package like the reference's
``beat_this.model``: ``beat_this_amd.model.beat_tracker.BeatThis`` and ``beat_this_amd.model.postprocessor.Postprocessor`` resolve
to the same classes, so ``pl_module.py``-style imports keep working.)  Precision
follows the caller exactly like the reference: under ``torch.autocast`` (what
``Spect2Frames(float16=True)`` enters, inference.py:246) the half-precision (fp16 MFMA operand) path runs, otherwise
an fp32-class path: exact fp32 MFMAs, or -- ``fp32_split_gemms``, what the inference classes select for
``float16=False`` -- three fp16 MFMAs per product on hi + lo operand halves.  There is no CPU implementation here.
"""
from __future__ import annotations

import weakref

import torch
from torch import nn

from .. import _lib
from ..pack import Engine, PackedModel
from ..weights import random_state_dict, resolve_hparams, state_dict_shapes

_BUFFER_LEAVES = ("running_mean", "running_var", "num_batches_tracked")
_TREE_EPOCH = [0]   # bumped whenever a module is assigned to / removed from a node of a BeatThis tree (_hooked_below's cache)


class _TracksChildren:
    """nn.Module mix-in: assigning, adding or deleting a sub-module bumps _TREE_EPOCH."""

    def __setattr__(self, name, value):
        if isinstance(value, nn.Module):
            _TREE_EPOCH[0] += 1
        super().__setattr__(name, value)

    def __delattr__(self, name):
        _TREE_EPOCH[0] += 1
        super().__delattr__(name)

    def add_module(self, name, module):
        _TREE_EPOCH[0] += 1
        super().add_module(name, module)


class _Node(_TracksChildren, nn.Module):
    """Container that carries the reference's dotted parameter names.  The nodes that are callable sub-modules in the
    reference -- ``frontend.stem``, ``.blocks``, ``.blocks[i]``, ``.blocks[i].partial``, ``.concat``, ``.linear``,
    ``transformer_blocks.layers[l][0]`` (Attention), ``[l][1]`` (FeedForward), ``.norm`` (beat_tracker.py:54-80,108-168,
    roformer.py:138-181) -- are bound to a unit of the owning model's engine (BeatThis._bind_units, bt_forward_unit) and can
    be called with the reference's tensor layouts: (b, c, f, t) inside the frontend, (b, n, dim) in the transformer; so are
    the four leaves of a partial transformer (``partial.attnF / .ffF / .attnT / .ffT``: (sequences, tokens, C), the branch
    without its residual, beat_tracker.py:251-301).  They run on the generic kernels (exact fp32, or half operands under
    autocast), one launch group per call; the fused fast path is ``BeatThis.forward`` / the three stages.  Nodes below them
    (``...norm``, ``.to_qkv``, ``.net`` ...) hold parameters only.
    Digit-named children index like the reference's ModuleList / Sequential (``layers[3][0]``, ``blocks[1]``)."""

    _unit = None   # (kind, index) once bound

    def __getstate__(self):  # (copy.deepcopy / pickle: the owner re-binds itself, BeatThis.__setstate__)
        state = self.__dict__.copy()
        state.pop("_root", None)
        return state

    def _indexed(self):
        keys = [k for k in self._modules if k.isdigit()]
        return sorted(keys, key=int)

    def __len__(self):
        return len(self._indexed())

    def __bool__(self):   # (a module is truthy whatever its number of indexed children)
        return True

    def __getitem__(self, i: int):
        keys = self._indexed()
        if not keys:
            raise TypeError(f"{type(self).__name__} node is not indexable")
        return self._modules[keys[i]]

    def __iter__(self):
        return iter(self._modules[k] for k in self._indexed())

    def forward(self, x: torch.Tensor):
        if self._unit is None:
            raise NotImplementedError(
                "this node only carries parameters under the reference's names; the callable sub-modules are frontend.stem / "
                ".blocks / .blocks[i] / .blocks[i].partial (and its attnF / ffF / attnT / ffT) / .concat / .linear, "
                "transformer_blocks.layers[l][0] / [l][1] / .norm and the three stages")
        return self._root()._run_unit(x, *self._unit)


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
        if self._stage < 2 and _hooked_below(self):
            # somebody hooked a sub-module of this stage: run it through the sub-modules, like the reference's containers
            # (beat_tracker.py:77-80, roformer.py:176-181), so that the hook fires
            if self._stage == 0:
                return self.linear(self.concat(self.blocks(self.stem(x))))
            for attn, ff in self.layers:
                x = attn(x) + x
                x = ff(x) + x
            return self.norm(x)
        out = self._root()._run(x, self._stage, self._stage)
        return {"beat": out[0], "downbeat": out[1]} if self._stage == 2 else out


def _hooked_below(node: nn.Module) -> bool:
    """A forward (pre-)hook is registered on a sub-module of ``node`` (not on ``node`` itself).  Asked on every forward and every
    single-file call: walking ``node.modules()`` costs 0.1 ms for final0's ~200 nodes, so the hook dictionaries of the tree
    (created once per module, mutated in place by ``register_forward_*hook``) are collected once and only looked at afterwards;
    the collection is redone when any node of a BeatThis tree had a sub-module assigned, added or removed (_TREE_EPOCH)."""
    epoch = _TREE_EPOCH[0]
    cache = node.__dict__.get("_bt_hook_dicts")
    if cache is None or cache[0] != epoch:
        cache = (epoch, [(m._forward_hooks, m._forward_pre_hooks) for m in node.modules() if m is not node])
        node.__dict__["_bt_hook_dicts"] = cache
    for fwd, pre in cache[1]:
        if fwd or pre:
            return True
    return False


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


class BeatThis(_TracksChildren, nn.Module):
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
        self._bind_units()
        self._engine = None
        # outside autocast: True = every product of the forward on three half MFMAs over hi + lo operand halves
        # (BT_PREC_F32X3: fp32-class results -- 1e-5 at the logits, identical beats -- at 16/3 of the fp32 matrix rate; operands
        # beyond the fp16 range of a hi half are detected and the batch is repeated on the exact path, Engine.forward_stages);
        # False = exact fp32 MFMAs.  True is the default of the module itself since round 5: load_model()
        # -- what hubconf.py exports as ``beat_this`` and what pl_module.py hands to split_predict_aggregate -- returns a
        # model on the same precision the inference classes run (a bfloat16 build has no hi + lo path: exact there).
        self.fp32_split_gemms = True
        self.eval()

    def __getstate__(self):
        state = self.__dict__.copy()
        state["_engine"] = None   # (a handle of the HIP library: rebuilt on first use)
        return state

    def __setstate__(self, state):
        super().__setstate__(state)
        for name in ("frontend", "transformer_blocks", "task_heads"):
            object.__setattr__(self._modules[name], "_root", weakref.ref(self))
        self._bind_units()

    def _bind_units(self) -> None:
        """Make the reference's callable sub-modules callable here (see _Node)."""
        def bind(node, kind, index=0):
            object.__setattr__(node, "_root", weakref.ref(self))
            node._unit = (kind, index)
        fr = self.frontend
        if "concat" not in fr._modules:   # (parameter-free in the reference: Rearrange("b c f t -> b t (c f)"))
            fr.add_module("concat", _Node())
        bind(fr.stem, "stem")
        bind(fr.blocks, "blocks")
        for i, blk in enumerate(fr.blocks):
            bind(blk, "block", i)
            if "partial" in blk._modules:
                bind(blk.partial, "partial", i)
                # its four leaves are ordinary Attention / FeedForward modules in the reference (beat_tracker.py:251-301)
                for j, (attn, ff) in enumerate((("attnF", "ffF"), ("attnT", "ffT"))):
                    bind(blk.partial._modules[attn], "fattn", 2 * i + j)
                    bind(blk.partial._modules[ff], "fff", 2 * i + j)
        bind(fr.concat, "concat")
        bind(fr.linear, "linear")
        for l, layer in enumerate(self.transformer_blocks.layers):
            bind(layer[0], "attn", l)
            bind(layer[1], "ff", l)
        bind(self.transformer_blocks.norm, "norm")

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
        if _hooked_below(self):
            # somebody hooked a sub-module: run the stages through the modules so the hooks fire (beat_tracker.py:188-192)
            return self.task_heads(self.transformer_blocks(self.frontend(x)))
        beat, down = self._run(x, 0, 2)
        return {"beat": beat, "downbeat": down}

    def _precision(self) -> int:
        """BT_PREC_* of a forward issued now: half under autocast, else hi + lo (fp32_split_gemms) or exact fp32."""
        half = torch.is_autocast_enabled("cuda") if hasattr(torch, "is_autocast_enabled") else False
        if half:
            return _lib.PREC_HALF
        return _lib.PREC_F32X3 if self.fp32_split_gemms and not _lib.lib().bt_half_is_bf16() else _lib.PREC_F32

    def _run(self, x: torch.Tensor, first: int, last: int, out=None):
        """Stages first..last in the engine; precision follows autocast like the whole forward.  ``out``: see
        Engine.forward_stages."""
        if x.dim() != 3:
            raise ValueError(f"expected a (batch, time, features) input, got {tuple(x.shape)}")
        _lib.require_gpu(x, "stage input")
        if x.shape[0] == 0 or x.shape[1] == 0:
            D = self.hparams["transformer_dim"]
            empty = torch.empty((x.shape[0], x.shape[1]), dtype=torch.float32, device=x.device)
            return (empty, empty.clone()) if last == 2 else torch.empty((x.shape[0], x.shape[1], D), dtype=torch.float32, device=x.device)
        return self.engine().forward_stages(x, self._precision(), first, last, logits_out=out)

    def _run_unit(self, x: torch.Tensor, kind: str, index: int):
        """A sub-module call (see _Node): reference layouts in and out, fp32 results; precision follows autocast (the hi + lo
        mode of ``fp32_split_gemms`` is the exact fp32 path here)."""
        _lib.require_gpu(x, "sub-module input")
        half = torch.is_autocast_enabled("cuda") if hasattr(torch, "is_autocast_enabled") else False
        prec = _lib.PREC_HALF if half else _lib.PREC_F32
        eng = self.engine()
        D = self.hparams["transformer_dim"]

        def to_btfc(t, i):   # (b, c, f, t) of frontend block i's input -> this library's (b, t, f, c)
            c, f = 32 << i, 32 >> i
            if t.dim() != 4 or t.shape[1] != c or t.shape[2] != f:
                raise ValueError(f"expected a (batch, {c}, {f}, time) input, got {tuple(t.shape)}")
            return t.permute(0, 3, 2, 1).contiguous()

        def partial(t, i):   # (b, t, f, c) -> (b, t, f, c)
            if not self.hparams["partial_transformers"]:
                return t
            return eng.forward_unit(t, prec, _lib.UNIT_PARTIAL, i, t.shape)

        def conv(t, i):      # (b, t, f, c) -> (b, t, f / 2, 2 c)
            b, n, f, c = t.shape
            return eng.forward_unit(t, prec, _lib.UNIT_CONV, i, (b, n, f // 2, 2 * c))

        if kind == "stem":
            if x.dim() != 3 or x.shape[2] != 128:
                raise ValueError(f"expected a (batch, time, 128) input, got {tuple(x.shape)}")
            return eng.forward_unit(x, prec, _lib.UNIT_STEM, 0, (x.shape[0], x.shape[1], 32, 32)).permute(0, 3, 2, 1)
        if kind == "partial":
            return partial(to_btfc(x, index), index).permute(0, 3, 2, 1)
        if kind == "block":
            blk = self.frontend.blocks[index]
            if _hooked_below(blk) and "partial" in blk._modules:   # (through the module: a hook on .partial fires)
                return conv(to_btfc(blk.partial(x), index), index).permute(0, 3, 2, 1)
            return conv(partial(to_btfc(x, index), index), index).permute(0, 3, 2, 1)
        if kind == "blocks":
            if _hooked_below(self.frontend.blocks):
                for blk in self.frontend.blocks:
                    x = blk(x)
                return x
            t = to_btfc(x, 0)
            for i in range(3):
                t = conv(partial(t, i), i)
            return t.permute(0, 3, 2, 1)
        if kind == "concat":
            if x.dim() != 4:
                raise ValueError(f"expected a (batch, channels, freq, time) input, got {tuple(x.shape)}")
            b, c, f, n = x.shape
            return x.permute(0, 3, 1, 2).reshape(b, n, c * f)
        if kind == "linear":
            if x.dim() != 3 or x.shape[2] != 1024:
                raise ValueError(f"expected a (batch, time, 1024) input, got {tuple(x.shape)}")
            b, n, _ = x.shape
            t = x.reshape(b, n, 256, 4).transpose(2, 3).contiguous()   # (c f) -> (f c): the packed weight's column order
            return eng.forward_unit(t, prec, _lib.UNIT_LINEAR, 0, (b, n, D))
        if kind in ("fattn", "fff"):
            # leaves of a PartialFTTransformer: Attention / FeedForward of width C on "(b t) f c" or "(b f) t c" rows; the branch
            # WITHOUT the residual like the main layers' (see below)
            c = 32 << (index >> 1)
            if x.dim() != 3 or x.shape[2] != c:
                raise ValueError(f"expected a (sequences, tokens, {c}) input, got {tuple(x.shape)}")
            xf = x.to(torch.float32).contiguous()
            y = eng.forward_unit(xf, prec, _lib.UNIT_FRONT_ATTN if kind == "fattn" else _lib.UNIT_FRONT_FF, index, xf.shape)
            return y - xf
        if x.dim() != 3 or x.shape[2] != D:
            raise ValueError(f"expected a (batch, time, {D}) input, got {tuple(x.shape)}")
        if kind == "norm":
            return eng.forward_unit(x, prec, _lib.UNIT_NORM, 0, x.shape)
        # Attention / FeedForward return their branch WITHOUT the residual (roformer.py:176-181 adds it); the engine computes
        # the residual form x + f(x) in fp32, so f(x) is that minus x (exact to the rounding of the fp32 sum)
        xf = x.to(torch.float32).contiguous()
        y = eng.forward_unit(xf, prec, _lib.UNIT_ATTN if kind == "attn" else _lib.UNIT_FF, index, xf.shape)
        return y - xf
