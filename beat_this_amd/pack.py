"""Pack a reference-layout BeatThis state dict into the kernel
layouts of include/beat_this_amd.h: fold RMSNorm gammas / BatchNorm statistics into the
adjacent weights, append the gate rows to the QKV projection, permute the conv and the
frontend.linear weights to the (b, t, f, c) activation layout, pad N to 128 rows and keep an
fp32 and a half-precision (fp16, or bf16 in a -DBT_HALF_BF16 build) device copy of every matrix.  One-time host work (torch CPU ops)."""
from __future__ import annotations

import collections
import ctypes as C
import math

import torch

from . import _lib, tables

LOG2E = 1.4426950408889634
BN_EPS = 1e-5


def perm32(w: torch.Tensor) -> torch.Tensor:
    """Reorder the columns inside every block of 32: new[16 g + r] = old[(r&3) + 8 (r>>2) + 4 g]
    (the MFMA 32x32 C-layout row order), see csrc/fused.hip."""
    idx = torch.tensor([(r & 3) + 8 * (r >> 2) + 4 * g for g in range(2) for r in range(16)])
    n, k = w.shape
    return w.reshape(n, k // 32, 32)[:, :, idx].reshape(n, k).contiguous()


def fragment_tiles(w: torch.Tensor) -> torch.Tensor:
    """[R, K] (R, K multiples of 32) -> [R/32, K/32, 64, 16]: per 32x32 tile, lane l = (row l & 31,
    16 contiguous k at 16 * (l >> 5)) -- the A/B operand fragment of one lane (csrc/common.h)."""
    r, k = w.shape
    t = w.reshape(r // 32, 32, k // 32, 2, 16)           # [rt, row, kt, g, i]
    return t.permute(0, 2, 3, 1, 4).reshape(r // 32, k // 32, 64, 16)  # lane = g * 32 + row


def ff_fragment_major(w1: torch.Tensor, w2p: torch.Tensor, elem_per_piece: int) -> torch.Tensor:
    """Fragment-major FF weights for ff_fused_kernel: per hidden block hb, KT tiles of W1 then KT tiles of
    PERM32'd W2; inside a tile the 16 elements of a lane are split into pieces of ``elem_per_piece``
    (8 for half = one 16-byte read, 4 for fp32) stored piece-major: [piece][lane][elem]."""
    dim = w1.shape[1]
    kt = dim // 32
    t1 = fragment_tiles(w1)                      # [HB, KT, 64, 16]
    t2 = fragment_tiles(w2p).permute(1, 0, 2, 3)  # [HB (col block), KT (row tile), 64, 16]
    blk = torch.cat([t1, t2], dim=1)             # [HB, 2 KT, 64, 16]
    pieces = 16 // elem_per_piece
    blk = blk.reshape(blk.shape[0], 2 * kt, 64, pieces, elem_per_piece).permute(0, 1, 3, 2, 4)
    return blk.contiguous().reshape(-1)


def qkv_fragment_major(w: torch.Tensor, dim: int) -> torch.Tensor:
    """half-precision QKV+gate weights for qkv_front_kernel: ``w`` = [3 dim + heads (padded to >= 3 dim + 32), dim];
    per head the k-tiles of its q, k, v row blocks, then the k-tiles of the gate row block; a tile is
    [half h][lane][8] (see fragment_tiles)."""
    heads = dim // 32
    t = fragment_tiles(w[:3 * dim + 32])                       # [3 heads + 1, KT, 64, 16]
    order = [blk * heads + hd for hd in range(heads) for blk in range(3)] + [3 * heads]
    t = t[order]                                               # [(hd, q|k|v)..., gate][KT][64][16]
    t = t.reshape(t.shape[0], t.shape[1], 64, 2, 8).permute(0, 1, 3, 2, 4)
    return t.contiguous().reshape(-1)


def tail_fragment_major(wo: torch.Tensor, w1p: torch.Tensor, w2p: torch.Tensor) -> torch.Tensor:
    """Weight stream of csrc/tail.hip (layout: bt_pair_weights.w_tail_frag): ``wo`` = to_out weight [C, C] (natural k
    order), ``w1p`` = gamma-folded net.1.weight [H, C] with PERM32'd columns, ``w2p`` = net.4.weight [C, H] with PERM32'd
    columns.  Every step is 2 C/32 fragment-major tiles ([half][lane][8])."""
    C = wo.shape[0]
    kt, hb = C // 32, w1p.shape[0] // 32

    def pieces(tiles):  # [..., 64, 16] -> [..., 2, 64, 8]
        return tiles.reshape(*tiles.shape[:-2], 64, 2, 8).transpose(-3, -2)

    ot = pieces(fragment_tiles(wo))       # [mt, kt, 2, 64, 8]
    t1 = pieces(fragment_tiles(w1p))      # [hb, kt, ...]
    t2 = pieces(fragment_tiles(w2p))      # [mt, hb, ...]
    zero = torch.zeros(kt * 1024, dtype=wo.dtype)
    steps = [ot[2 * st: 2 * st + 2].reshape(-1) for st in range(kt // 2)]
    for i in range(-1, hb + 1):
        steps.append(t1[i + 1].reshape(-1) if i + 1 < hb else zero)
        steps.append(t2[:, i - 1].reshape(-1) if i >= 1 else zero)
    return torch.cat(steps)


def _pad_rows(w: torch.Tensor, mult: int = 128) -> torch.Tensor:
    n = w.shape[0]
    n_pad = (n + mult - 1) // mult * mult
    if n_pad == n:
        return w.contiguous()
    out = torch.zeros((n_pad, w.shape[1]), dtype=w.dtype)
    out[:n] = w
    return out


class PackedModel:
    """Owns the device tensors and the bt_model_desc that points into them."""

    def __init__(self, state_dict: dict, hparams: dict, device: torch.device):
        sd = {k: v.detach().to("cpu", torch.float32) if v.is_floating_point() else v.detach().cpu()
              for k, v in state_dict.items()}
        self.device = torch.device(device)
        self._keep: list[torch.Tensor] = []
        self._hl8_todo: list = []    # (pair weights, hl8 field, fp32 field, rows) of the main layers' GEMM weights: ensure_hl8
        self._by_ptr: dict = {}      # data_ptr -> fp32 device copy written by _mat
        self.desc = _lib.ModelDesc()
        d = self.desc
        D = int(hparams["transformer_dim"])
        L = int(hparams["n_layers"])
        if int(hparams.get("head_dim", 32)) != 32 or int(hparams.get("stem_dim", 32)) != 32 or \
                int(hparams.get("spect_dim", 128)) != 128:
            raise ValueError("beat_this_amd kernels are built for head_dim=32, stem_dim=32, spect_dim=128")
        mult = int(hparams.get("ff_mult", 4))
        if not 1 <= mult <= 16 or (mult * D) % 128:
            raise ValueError(f"unsupported ff_mult={mult} for transformer_dim={D} (ff_mult * transformer_dim must be a "
                             "multiple of 128)")
        d.ff_mult = mult
        if L > _lib.MAX_LAYERS or D % 32:
            raise ValueError(f"unsupported transformer_dim={D} / n_layers={L}")
        d.transformer_dim, d.n_layers = D, L
        d.sum_head = int(bool(hparams.get("sum_head", True)))
        d.partial_transformers = int(bool(hparams.get("partial_transformers", True)))

        # ---- stem: BN1d -> scale/shift, BN2d folded into the conv ------------------------
        s1 = sd["frontend.stem.bn1d.weight"] / torch.sqrt(sd["frontend.stem.bn1d.running_var"] + BN_EPS)
        d.bn1_scale = self._f32(s1)
        d.bn1_shift = self._f32(sd["frontend.stem.bn1d.bias"] - sd["frontend.stem.bn1d.running_mean"] * s1)
        s2 = sd["frontend.stem.bn2d.weight"] / torch.sqrt(sd["frontend.stem.bn2d.running_var"] + BN_EPS)
        d.stem_w = self._f32(sd["frontend.stem.conv2d.weight"].reshape(32, 12) * s2[:, None])  # [co][df*3+dt]
        d.stem_b = self._f32(sd["frontend.stem.bn2d.bias"] - sd["frontend.stem.bn2d.running_mean"] * s2)

        # ---- frontend blocks ---------------------------------------------------------------
        dim = 32
        for i in range(3):
            p = f"frontend.blocks.{i}."
            if d.partial_transformers:
                self._pair(d.front[i][0], sd, p + "partial.attnF.", p + "partial.ffF.", dim)
                self._pair(d.front[i][1], sd, p + "partial.attnT.", p + "partial.ffT.", dim)
            sc = sd[p + "norm.weight"] / torch.sqrt(sd[p + "norm.running_var"] + BN_EPS)
            w = sd[p + "conv2d.weight"] * sc[:, None, None, None]           # [co, ci, df, dt]
            w = w.permute(0, 3, 2, 1).reshape(2 * dim, 6 * dim)             # [co][dt][df][ci]
            d.conv_w[i][0], d.conv_w[i][1] = self._mat(w)
            if 2 * dim >= 128:   # (the first convolution, N = 64, stays on the register-staged GEMM)
                d.conv_w_x3[i] = self._hl32(w)
            d.conv_b[i] = self._f32(sd[p + "norm.bias"] - sd[p + "norm.running_mean"] * sc)
            dim *= 2
        # ---- frontend.linear: reference column = c*4 + f, ours = f*256 + c -----------------
        w = sd["frontend.linear.weight"].view(D, dim, 4).permute(0, 2, 1).reshape(D, 4 * dim)
        d.lin_w[0], d.lin_w[1] = self._mat(w)
        d.lin_w_x3 = self._hl32(w)
        d.lin_b = self._f32(sd["frontend.linear.bias"])
        # ---- transformer layers ----------------------------------------------------------------
        for l in range(L):
            p = f"transformer_blocks.layers.{l}."
            self._pair(d.layers[l], sd, p + "0.", p + "1.", D, x3=True)
        g = sd["transformer_blocks.norm.gamma"]
        d.head_w = self._f32(sd["task_heads.beat_downbeat_lin.weight"] * g[None, :])
        d.norm_out_g, d.head_w_raw = self._f32(g), self._f32(sd["task_heads.beat_downbeat_lin.weight"])  # (stage calls)
        hb = sd["task_heads.beat_downbeat_lin.bias"]
        d.head_b[0], d.head_b[1] = float(hb[0]), float(hb[1])
        self._freqs = next(v for k, v in sd.items() if k.endswith("rotary_embed.freqs"))
        self.set_positions(1536)

    def set_positions(self, n_pos: int) -> None:
        """(Re)build the rotary table for sequences of up to ``n_pos`` frames (bt_model_desc.rope / rope_len); a previous
        table is released (the caller makes sure nothing still reads it: Engine.ensure_positions)."""
        old = getattr(self, "_rope_t", None)
        self._rope_t = torch.from_numpy(tables.rope_table(self._freqs, n_pos)).to(torch.float32).contiguous().to(self.device)
        self.desc.rope = self._rope_t.data_ptr()
        self.desc.rope_len = int(n_pos)
        del old

    # ------------------------------------------------------------------------------------------
    def _f32(self, t: torch.Tensor) -> int:
        t = t.to(torch.float32).contiguous().to(self.device)
        self._keep.append(t)
        return t.data_ptr()

    def _x3_stream(self, flat: torch.Tensor) -> int:
        """BT_PREC_F32X3 form of a half fragment stream (fp32 values in half-fragment order, whole 32 x 32 tiles): the
        weights times 64 (csrc/common.h: OpScale -- keeps the lo halves normal fp16 numbers), per tile the hi half tile
        followed by the lo half tile (64 w = hi + lo to 2^-22)."""
        flat = flat * 64.0
        hi = flat.to(torch.float16)
        lo = (flat - hi.to(torch.float32)).to(torch.float16)
        t = torch.cat([hi.reshape(-1, 1024), lo.reshape(-1, 1024)], 1).reshape(-1).to(self.device)
        self._keep.append(t)
        return t.data_ptr()

    def _mat(self, w: torch.Tensor):
        """[N padded to 128][K] fp32 and half copies of a GEMM weight; the half copy is followed by its lo part
        (w - half(w), again in half): BT_PREC_F32X3 multiplies hi + lo, the half path reads the hi part only."""
        w = _pad_rows(w.to(torch.float32))
        a = w.to(self.device)
        ht = _lib.half_torch_dtype()
        hi = w.to(ht)
        b = torch.cat([hi, (w - hi.to(torch.float32)).to(ht)]).to(self.device)
        self._keep += [a, b]
        if hasattr(self, "_by_ptr"):
            self._by_ptr[a.data_ptr()] = a
        return a.data_ptr(), b.data_ptr()

    def _hl32(self, w: torch.Tensor) -> int:
        """BT_PREC_F32X3 form of a GEMM weight for csrc/gemm3.hip: fp32 [N, K] (K % 32 == 0) -> fp16 [N padded to 256][2 K],
        per 32 consecutive columns the 32 hi halves (half(w)) followed by the 32 lo halves (half(w - hi)): w = hi + lo to
        2^-22 (relative) for |w| >= 2^-3 and to 2^-25 absolute below.  0 (NULL) in a bfloat16 build."""
        if _lib.lib().bt_half_is_bf16():
            return 0
        w = _pad_rows(w.to(torch.float32), 256)
        hi = w.to(torch.float16)
        lo = (w - hi.to(torch.float32)).to(torch.float16)
        n, k = w.shape
        t = torch.stack([hi.reshape(n, k // 32, 32), lo.reshape(n, k // 32, 32)], 2).reshape(n, 2 * k).contiguous().to(self.device)
        self._keep.append(t)
        return t.data_ptr()

    def _hl8(self, w: torch.Tensor) -> int:
        """BT_OPT_X3_GEMM_FP8 form of a GEMM weight for csrc/gemm3.hip (X3 = 2): fp32 [N, K] (K % 32 == 0) -> [N padded to 256]
        rows of 128 B per 32 columns like ``_hl32``: the 32 hi halves (half(w)), the 32 hi bytes (e4m3(w)) and the 32 lo bytes
        (e4m3(2^11 (w - hi))) -- the two cross terms of the hi + lo product then run on one block-scaled fp8 MFMA per 32-k
        step.  0 (NULL) in a bfloat16 build or where torch has no float8_e4m3fn."""
        if _lib.lib().bt_half_is_bf16() or not hasattr(torch, "float8_e4m3fn"):
            return 0
        w = _pad_rows(w.to(torch.float32), 256)
        n, k = w.shape
        hi = w.to(torch.float16)
        lo = (w - hi.to(torch.float32)) * 2048.0
        t = torch.empty((n, k // 32, 128), dtype=torch.uint8)
        t[:, :, :64] = hi.contiguous().view(n, k // 32, 32).view(torch.uint8)
        t[:, :, 64:96] = w.clamp(-448, 448).to(torch.float8_e4m3fn).view(torch.uint8).view(n, k // 32, 32)
        t[:, :, 96:] = lo.clamp(-448, 448).to(torch.float8_e4m3fn).view(torch.uint8).view(n, k // 32, 32)
        t = t.contiguous().to(self.device)
        self._keep.append(t)
        return t.data_ptr()

    def ensure_hl8(self) -> bool:
        """Pack the hl8 (BT_OPT_X3_GEMM_FP8) forms of the main layers' GEMM weights if that has not happened yet; -> True when
        bt_pair_weights fields changed (a bt_engine copies its description at creation: the caller re-creates the handle)."""
        todo, self._hl8_todo = self._hl8_todo, []
        for pw, field, src, rows in todo:
            w = self._by_ptr[getattr(pw, src)[0]][:rows].cpu()
            setattr(pw, field, self._hl8(w))
        return bool(todo)

    def _pair(self, pw, sd, pa: str, pf: str, dim: int, x3: bool = False) -> None:
        heads = dim // 32
        pw.dim, pw.heads = dim, heads
        ga = sd[pa + "norm.gamma"]
        wqkv = sd[pa + "to_qkv.weight"].clone()            # rows (qkv h d)
        wqkv[:dim] *= LOG2E / math.sqrt(32.0)              # softmax scale + exp2 domain into q
        wg = sd[pa + "to_gates.weight"]
        w = torch.cat([wqkv, wg], 0) * ga[None, :]
        pw.w_qkvg[0], pw.w_qkvg[1] = self._mat(w)
        if dim <= 128:
            qflat = qkv_fragment_major(_pad_rows(w.to(torch.float32)), dim)
            qf = qflat.to(_lib.half_torch_dtype()).to(self.device)
            self._keep.append(qf)
            pw.w_qkv_frag = qf.data_ptr()
            if not _lib.lib().bt_half_is_bf16():
                pw.w_qkv_frag_x3 = self._x3_stream(qflat)
        pw.b_gates = self._f32(sd[pa + "to_gates.bias"])
        pw.w_out[0], pw.w_out[1] = self._mat(sd[pa + "to_out.0.weight"])
        if x3 and dim % 128 == 0:   # main layers: hl32 weights of the LDS-DMA GEMM (csrc/gemm3.hip, BT_PREC_F32X3)
            pw.w_qkvg_x3 = self._hl32(w)
            pw.w_out_x3 = self._hl32(sd[pa + "to_out.0.weight"])
            pw.w_ff1_x3 = self._hl32(sd[pf + "net.1.weight"] * sd[pf + "net.0.gamma"][None, :])
            pw.w_ff2_x3 = self._hl32(sd[pf + "net.4.weight"])
            # BT_OPT_X3_GEMM_FP8 (opt-in, off by default): the same four matrices with the lo halves as (hi byte, lo byte) pairs are
            # packed LAZILY, when the option is first switched on (ensure_hl8: from the fp32 device copies _mat keeps anyway) --
            # packed unconditionally they doubled the x3 weight memory of every engine for an option almost nobody sets
            if hasattr(self, "_hl8_todo"):
                self._hl8_todo += [(pw, "w_qkvg_f8", "w_qkvg", w.shape[0]), (pw, "w_out_f8", "w_out", dim),
                                   (pw, "w_ff1_f8", "w_ff1", sd[pf + "net.1.weight"].shape[0]), (pw, "w_ff2_f8", "w_ff2", dim)]
        # (the register-chained kernels of fused.hip / fused2.hip are built for a hidden width of 4 dim: a main layer with
        # another ff_mult and dim <= 128 runs on the plain GEMM path)
        if dim <= 128 and sd[pf + "net.1.weight"].shape[0] == 4 * dim:
            w1 = (sd[pf + "net.1.weight"] * sd[pf + "net.0.gamma"][None, :]).to(torch.float32)
            w2p = perm32(sd[pf + "net.4.weight"].to(torch.float32))
            f32 = ff_fragment_major(w1, w2p, 4).to(self.device)
            b16 = ff_fragment_major(w1, w2p, 8).to(_lib.half_torch_dtype()).to(self.device)
            self._keep += [f32, b16]
            pw.w_ff_frag[0], pw.w_ff_frag[1] = f32.data_ptr(), b16.data_ptr()
            # fused2.hip: out-projection tiles (natural k order), then the FF stream with PERM32'd W1 columns
            wo = sd[pa + "to_out.0.weight"].to(torch.float32)
            w1p = perm32(w1)
            for i, (epp, dt) in enumerate(((4, torch.float32), (8, _lib.half_torch_dtype()))):
                ot = fragment_tiles(wo)                                   # [mt, kt, 64, 16]
                ot = ot.reshape(ot.shape[0], ot.shape[1], 64, 16 // epp, epp).permute(0, 1, 3, 2, 4)
                ot = torch.cat([ot, torch.zeros_like(ot)], 1).reshape(-1)  # every step is 2 KT tiles: pad with zeros
                flat = torch.cat([ot, ff_fragment_major(w1p, w2p, epp)])
                t = flat.to(dt).to(self.device)
                self._keep.append(t)
                pw.w_outff_frag[i] = t.data_ptr()
                if i == 1 and dt == torch.float16:
                    pw.w_outff_frag_x3 = self._x3_stream(flat)
                # fused frequency-direction half: [gates | pad], per head [q | k] [v | outp tiles], FF steps
                kt = dim // 32

                def pieces(tiles):  # [..., 64, 16] -> piece-major tiles, flattened
                    sh = tiles.shape[:-2]
                    return tiles.reshape(*sh, 64, 16 // epp, epp).transpose(-3, -2).reshape(-1)
                qt = fragment_tiles(_pad_rows(w.to(torch.float32))[:3 * dim + 32])   # [3 heads + 1, KT, 64, 16]
                opt = fragment_tiles(perm32(wo))                                    # [mt, kt (head), 64, 16]
                steps = [pieces(qt[3 * heads]), torch.zeros(kt * 32 * 32)]
                for hd in range(heads):
                    steps += [pieces(qt[hd]), pieces(qt[heads + hd]), pieces(qt[2 * heads + hd]), pieces(opt[:, hd])]
                steps.append(ff_fragment_major(w1p, w2p, epp))
                flat = torch.cat(steps)
                t = flat.to(dt).to(self.device)
                self._keep.append(t)
                pw.w_attnff_frag[i] = t.data_ptr()
                if i == 1 and dt == torch.float16:
                    pw.w_attnff_frag_x3 = self._x3_stream(flat)
        gf = sd[pf + "net.0.gamma"]
        pw.w_ff1[0], pw.w_ff1[1] = self._mat(sd[pf + "net.1.weight"] * gf[None, :])
        pw.b_ff1 = self._f32(sd[pf + "net.1.bias"])
        pw.w_ff2[0], pw.w_ff2[1] = self._mat(sd[pf + "net.4.weight"])
        pw.b_ff2 = self._f32(sd[pf + "net.4.bias"])
        hidden = sd[pf + "net.1.weight"].shape[0]
        if dim in (256, 512) and hidden % 64 == 0 and 128 <= hidden <= 4096:  # fused layer tail (csrc/tail.hip)
            w1 = (sd[pf + "net.1.weight"] * gf[None, :]).to(torch.float32)
            t = tail_fragment_major(sd[pa + "to_out.0.weight"].to(torch.float32), perm32(w1),
                                    perm32(sd[pf + "net.4.weight"].to(torch.float32)))
            t = t.to(_lib.half_torch_dtype()).to(self.device)
            self._keep.append(t)
            pw.w_tail_frag = t.data_ptr()


class PackedPair:
    """One attention+FF pair packed on its own (single-operator tests and tools)."""

    def __init__(self, sd: dict, attn_prefix: str, ff_prefix: str, dim: int, device):
        self.device = torch.device(device)
        self._keep: list[torch.Tensor] = []
        self.weights = _lib.PairWeights()
        sd = {k: v.detach().to("cpu", torch.float32) for k, v in sd.items()}
        PackedModel._pair(self, self.weights, sd, attn_prefix, ff_prefix, dim, x3=True)

    _f32 = PackedModel._f32
    _mat = PackedModel._mat
    _hl32 = PackedModel._hl32
    _hl8 = PackedModel._hl8
    _x3_stream = PackedModel._x3_stream


class Engine:
    """bt_engine handle + a growable workspace on one device."""

    def __init__(self, packed: PackedModel):
        self.packed = packed
        self.device = packed.device
        h = C.c_void_p()
        _lib.check(_lib.lib().bt_engine_create(C.byref(packed.desc), C.byref(h)))
        self._h = h
        # One workspace per stream (the C ABI's workspace belongs to ONE stream at a time), keyed by the torch Stream
        # (torch hands out its streams from a fixed pool, so equal keys are the same HIP stream, and a workspace is
        # allocated under the stream that uses it: the caching allocator's stream-ordered reuse covers its release).  At
        # most MAX_WORKSPACES are kept (least recently used goes first: callers that cycle through many streams do not
        # leak one workspace each), and a workspace far larger than what the stream's recent calls need (> 4x, SHRINK_AFTER
        # times in a row) is dropped instead of pinning e.g. 6.7 GB of a past 96-chunk fp32 batch for the engine's lifetime.
        # Footprint: ~66 MB (fp32) / ~45 MB (half) per chunk and stream; inference.py runs slices of <= 96 chunks
        # (MAX_CHUNKS_PER_LAUNCH) on the caller's stream + CONCURRENT_STREAMS side streams.
        self._ws = collections.OrderedDict()
        self._deferred = None      # list collecting (pinned flag, event) pairs while deferred_range_checks() is active
        self.last_fallbacks = 0    # BT_PREC_F32X3 forwards whose range flag fired (and were repeated in exact fp32)

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                _lib.lib().bt_engine_destroy(self._h)
                self._h = None
        except Exception:
            pass

    MAX_WORKSPACES = 4
    SHRINK_AFTER = 8   # consecutive requests of < 1/4 of a stream's workspace before it is given back
    OPTIONS = {"x3_attn_p16": 1, "x3_gemm_fp8": 2}   # name -> BT_OPT_* of include/beat_this_amd.h

    def set_options(self, opts: dict) -> None:
        """Arithmetic variants of the engine (bt_engine_set_option), e.g. ``{"x3_attn_p16": 0}`` for the three-term P.V of
        rounds 3 - 4 everywhere, ``1`` (default) for P16 in the main layers only, ``2`` main layers + frontend (round 5's default);
        ``{"x3_gemm_fp8": 1 | 2}`` for BASELINE config 5 (fp8 cross terms in the feed-forward / in all main-layer GEMMs).  Captured forwards are dropped: a graph replays the kernels it was recorded with."""
        for name in opts:
            if name not in self.OPTIONS:
                raise ValueError(f"unknown engine option {name!r} (known: {sorted(self.OPTIONS)})")
        if int(opts.get("x3_gemm_fp8", 0)) > 0 and self.packed.ensure_hl8():
            self._recreate_handle()   # (the hl8 weights were packed just now: the handle's copy of the description is stale)
        for name, value in opts.items():
            _lib.check(_lib.lib().bt_engine_set_option(self._h, self.OPTIONS[name], int(value)))
        self._drop_graphs()

    def _drop_graphs(self) -> None:
        """Captured forwards replay the kernels / tables they were recorded with: dropped together with their per-stream
        workspaces (up to ~0.9 GB for 11 chunks), which nothing else refers to."""
        self.__dict__.pop("_graphs", None)
        self.__dict__.pop("_graph_ws", None)

    def _recreate_handle(self) -> None:
        """A new bt_engine from the (changed) description, options carried over; the old handle is released afterwards."""
        if self._deferred is not None:
            raise RuntimeError("the engine cannot be re-created while range checks of earlier forwards are pending")
        torch.cuda.synchronize(self.device)
        h = C.c_void_p()
        _lib.check(_lib.lib().bt_engine_create(C.byref(self.packed.desc), C.byref(h)))
        for name in self.OPTIONS:
            _lib.check(_lib.lib().bt_engine_set_option(h, self.OPTIONS[name], self.get_option(name)))
        old_h, self._h = self._h, h
        _lib.lib().bt_engine_destroy(old_h)
        self._drop_graphs()

    def get_option(self, name: str) -> int:
        v = C.c_int(0)
        _lib.check(_lib.lib().bt_engine_get_option(self._h, self.OPTIONS[name], C.byref(v)))
        return int(v.value)

    def ensure_positions(self, T: int) -> None:
        """Sequences longer than the rotary table (1536 rows by default: the reference's chunks are 1500 frames, but its module
        takes any length, beat_tracker.py:188-192): grow the table to the next multiple of 512 and re-create the handle."""
        if T <= self.packed.desc.rope_len:
            return
        if self._deferred is not None:
            raise RuntimeError("the rotary table cannot grow while range checks of earlier forwards are pending")
        torch.cuda.synchronize(self.device)   # (nothing may still be reading the old table when it is released)
        old_rope, old_len, old_t = self.packed.desc.rope, self.packed.desc.rope_len, self.packed._rope_t
        self.packed.set_positions(-(-T // 512) * 512)
        h = C.c_void_p()
        try:
            _lib.check(_lib.lib().bt_engine_create(C.byref(self.packed.desc), C.byref(h)))
        except Exception:   # the old handle (and its table) stay valid
            self.packed.desc.rope, self.packed.desc.rope_len, self.packed._rope_t = old_rope, old_len, old_t
            raise
        self._drop_graphs()   # (captured forwards hold the old table's address)
        for name in self.OPTIONS:            # the new handle inherits the old one's options
            _lib.check(_lib.lib().bt_engine_set_option(h, self.OPTIONS[name], self.get_option(name)))
        old_h, self._h = self._h, h       # swap first, then release: self._h never dangles
        _lib.lib().bt_engine_destroy(old_h)   # (its profiling records, if a bench leg had some open, go with it)

    def _workspace(self, need: int) -> torch.Tensor:
        key = torch.cuda.current_stream(self.device)
        ws, small = self._ws.pop(key, (None, 0))
        if ws is not None and ws.numel() < need:
            ws = None
        elif ws is not None and ws.numel() > 4 * need:
            # far larger than this call needs: kept while big and small batches alternate on the stream (a 96-chunk slice
            # next to short pieces would otherwise free and re-allocate gigabytes every call), dropped after
            # SHRINK_AFTER consecutive small requests
            small += 1
            if small >= self.SHRINK_AFTER:
                ws = None
        else:
            small = 0
        if ws is None:
            ws, small = torch.empty(need, dtype=torch.uint8, device=self.device), 0
        self._ws[key] = (ws, small)   # (most recently used last)
        while len(self._ws) > self.MAX_WORKSPACES:
            self._ws.popitem(last=False)
        return ws

    def forward(self, spect: torch.Tensor, prec: int):
        """spect: (B, T, 128) fp32 on the engine's device -> (beat, downbeat) fp32 (B, T)."""
        return self.forward_stages(spect, prec, 0, 2)

    def forward_stages(self, spect: torch.Tensor, prec: int, first: int, last: int, logits_out=None):
        """Stages first..last of BeatThis.forward (0 frontend, 1 transformer_blocks, 2 task_heads; bt_forward_stages):
        (B, T, 128) or (B, T, D) fp32 in -> (B, T, D) fp32, or (beat, downbeat) when the head is included.
        ``logits_out`` = (beat, downbeat): contiguous fp32 (B, T) tensors the logits are written into (last == 2 only)."""
        _lib.require_gpu(spect, "stage input")
        B, T, M = spect.shape
        D = self.packed.desc.transformer_dim
        if M != (128 if first == 0 else D):
            raise ValueError(f"expected {128 if first == 0 else D} input features, got {M}")
        x = spect.to(torch.float32).contiguous()
        self.ensure_positions(T)
        need = _lib.lib().bt_workspace_bytes(self._h, B, T, prec)
        if need == 0:
            raise ValueError("empty batch")
        ws = self._workspace(need)
        beat = down = out = None
        if last == 2 and logits_out is not None:
            beat, down = logits_out
            for t in (beat, down):
                if t.dtype != torch.float32 or tuple(t.shape) != (B, T) or not t.is_contiguous() or t.device != x.device:
                    raise ValueError("logits_out: two contiguous fp32 (batch, time) tensors on the input's device")
        elif last == 2:
            beat = torch.empty((B, T), dtype=torch.float32, device=self.device)
            down = torch.empty((B, T), dtype=torch.float32, device=self.device)
        else:
            out = torch.empty((B, T, D), dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().bt_forward_stages(self._h, _lib.stream_ptr(self.device), prec, first, last, x.data_ptr(), B, T,
                                                    ws.data_ptr(), ws.numel(), _lib.ptr(out), _lib.ptr(beat), _lib.ptr(down)))
            if prec == _lib.PREC_F32X3:
                # Range guard of the hi + lo split (include/beat_this_amd.h, BT_PREC_F32X3): the first word of the workspace is
                # non-zero when an operand left the fp16 range of a hi part (the result then holds inf / NaN).  Default: look at
                # it now (one stream synchronisation per forward) and repeat the batch on the exact fp32 MFMA path.  A caller
                # that pipelines several forwards collects the flags instead (deferred_range_checks) and decides later.
                flag = ws[:4].view(torch.int32)
                if self._deferred is not None:
                    host = torch.empty(1, dtype=torch.int32, pin_memory=True)
                    host.copy_(flag, non_blocking=True)
                    ev = torch.cuda.Event()
                    ev.record(torch.cuda.current_stream(self.device))
                    self._deferred.append((host, ev))
                elif int(flag.item()) != 0:
                    self.last_fallbacks += 1
                    return self.forward_stages(spect, _lib.PREC_F32, first, last, logits_out=logits_out)
        return (beat, down) if last == 2 else out

    def forward_unit(self, x: torch.Tensor, prec: int, unit: int, index: int, out_shape) -> torch.Tensor:
        """One sub-module of the model (bt_forward_unit; the _lib.UNIT_* constants): fp32 (B, T, ...) tensor in this
        library's activation layout in, a new fp32 tensor of ``out_shape`` out.  Generic kernels, exact fp32 or half operands."""
        _lib.require_gpu(x, "sub-module input")
        B, T = int(x.shape[0]), int(x.shape[1])
        if B == 0 or T == 0:
            raise ValueError("empty batch")
        if prec == _lib.PREC_F32X3:
            prec = _lib.PREC_F32
        x = x.to(torch.float32).contiguous()
        self.ensure_positions(T)
        ws = self._workspace(_lib.lib().bt_workspace_bytes(self._h, B, T, prec))
        out = torch.empty(tuple(out_shape), dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().bt_forward_unit(self._h, _lib.stream_ptr(self.device), prec, unit, index, x.data_ptr(),
                                                  out.data_ptr(), B, T, ws.data_ptr(), ws.numel()))
        return out

    # -- small batches as hipGraphs -----------------------------------------------------------------------------------------
    GRAPH_MAX_CHUNKS = 11    # a 5-minute track; beyond that the launches of a forward are a negligible share of it
    GRAPH_T = 1500           # only full-length chunks are captured: B is then the only variable (pieces of <= 1488 frames
                             # have T = frames + 12 -- one capture per clip length would cost more than it ever saves)
    GRAPH_MAX_ENTRIES = 12   # graphs kept per engine (least recently used goes first); entries of a stream share a workspace

    def graph_forward(self, B: int, T: int, prec: int):
        """The whole forward of a (B, T, 128) batch as ONE hipGraph, for the single-file path (a 30 s file is 2 chunks = ~55
        launches whose host side is a sixth of the call): -> entry with ``.x`` (B, T, 128) to fill, ``.replay()`` and
        ``.beat`` / ``.down`` (B, T) holding the logits afterwards (valid until the next replay on the same stream), or None
        where graphs are off / not applicable (T != GRAPH_T, too many chunks, pending range checks, profiling).  Entries are
        per (stream, B, precision); a capture that fails -- or whose replay does not reproduce the plain launches -- switches
        graphs off for this engine and the caller falls back to plain launches."""
        if not getattr(self, "_graphs_ok", True) or T != self.GRAPH_T or B > self.GRAPH_MAX_CHUNKS or self._deferred is not None \
                or self._h_prof_on():
            return None
        stream = torch.cuda.current_stream(self.device)
        key = (stream, B, T, prec)
        e = self.__dict__.setdefault("_graphs", collections.OrderedDict()).pop(key, None)
        if e is None:
            try:
                e = _GraphEntry(self, B, T, prec, stream)
            except Exception as err:  # noqa: BLE001  (capture not available: plain launches from now on)
                self._graphs_ok = False
                self._graph_error = repr(err)
                return None
        # (the dictionary again: constructing an entry may have grown the rotary table, which drops the captured forwards)
        cache = self.__dict__.setdefault("_graphs", collections.OrderedDict())
        cache[key] = e
        while len(cache) > self.GRAPH_MAX_ENTRIES:
            cache.popitem(last=False)
        return e

    def _graph_workspace(self, stream, need: int) -> torch.Tensor:
        """One workspace per stream for all captured forwards replayed on it (replays of one stream run in order, so they can
        share it); an entry that needs more gets a larger one, which later entries share (older graphs keep theirs alive)."""
        pool = self.__dict__.setdefault("_graph_ws", {})
        ws = pool.get(stream)
        if ws is None or ws.numel() < need:
            ws = pool[stream] = torch.empty(need, dtype=torch.uint8, device=self.device)
        return ws

    def _a2b_workspace(self, need: int) -> torch.Tensor:
        """Device scratch of Audio2Beats' one-call path (bt_audio2beats_enqueue), one per stream, grown in steps: the forward
        graph the library captures is keyed by this allocation, so it stays put from call to call (a longer file than any
        before re-allocates -- and re-captures -- once)."""
        stream = torch.cuda.current_stream(self.device)
        pool = self.__dict__.setdefault("_a2b_ws", {})
        ws, small = pool.get(stream, (None, 0))
        if ws is not None and ws.numel() > 4 * need:
            # far larger than this call needs (one very long file among short ones: ~80 MB per chunk): kept for a while, given
            # back after SHRINK_AFTER consecutive small requests -- the same policy as the forward's workspaces
            small += 1
            if small >= self.SHRINK_AFTER:
                ws = None
        else:
            small = 0
        if ws is None or ws.numel() < need:
            ws, small = torch.empty(int(need * 1.25) if ws is not None else need, dtype=torch.uint8, device=self.device), 0
        pool[stream] = (ws, small)
        while len(pool) > self.MAX_WORKSPACES:
            pool.pop(next(iter(pool)))
        return ws

    def _h_prof_on(self) -> bool:
        return bool(getattr(self, "profiling", False))

    # -- BT_PREC_F32X3 range guard, deferred form ------------------------------------------------------------------------
    def deferred_range_checks(self):
        """Context manager: BT_PREC_F32X3 forwards inside it do not synchronise; their range flags are copied to pinned host
        memory asynchronously and collected in the list the context yields.  ``Engine.range_exceeded(list)`` (after the
        work has been waited for, or waiting itself) tells whether any of them fired -- the caller then repeats that
        work with the model's ``fp32_split_gemms`` off."""
        import contextlib

        @contextlib.contextmanager
        def ctx():
            outer, mine = self._deferred, []
            self._deferred = mine
            try:
                yield mine
            finally:
                self._deferred = outer
                if outer is not None:
                    outer.extend(mine)
        return ctx()

    def range_exceeded(self, checks) -> bool:
        bad = False
        for host, ev in checks:
            ev.synchronize()
            bad = bad or int(host[0]) != 0
        if bad:
            self.last_fallbacks += 1
        return bad


class _GraphEntry:
    """One captured forward (Engine.graph_forward)."""

    def __init__(self, eng: Engine, B: int, T: int, prec: int, stream):
        dev = eng.device
        eng.ensure_positions(T)
        self.eng, self.B, self.T, self.prec = eng, B, T, prec
        # Everything below happens with the ENGINE's device current and on streams of that device: torch.cuda.graph's default
        # capture stream belongs to whatever device was current when it was first made, and kernels launched on another
        # stream than the capturing one run eagerly and leave the graph empty (a model on cuda:1 while cuda:0 is
        # current replayed nothing and returned the logits of the warm-up run).
        with torch.cuda.device(dev):
            self.x = torch.zeros((B, T, 128), dtype=torch.float32, device=dev)
            self.beat = torch.empty((B, T), dtype=torch.float32, device=dev)
            self.down = torch.empty((B, T), dtype=torch.float32, device=dev)
            self.ws = eng._graph_workspace(stream, _lib.lib().bt_workspace_bytes(eng._h, B, T, prec))
            self.flag = self.ws[:4].view(torch.int32)
            self._launch()                       # once eagerly (lazy module loading, allocator warm-up) ...
            torch.cuda.synchronize(dev)
            want = (self.beat.clone(), self.down.clone())
            capture = torch.cuda.Stream(device=dev)
            capture.wait_stream(torch.cuda.current_stream(dev))
            self.graph = torch.cuda.CUDAGraph()
            # (thread_local: other host threads keep launching on their own streams while this one records)
            with torch.cuda.graph(self.graph, stream=capture, capture_error_mode="thread_local"):   # ... then recorded on a stream of the engine's device
                self._launch()
            # a capture is only trusted once a replay has reproduced the plain launches bit for bit (same kernels, same input)
            self.beat.fill_(float("nan"))
            self.down.fill_(float("nan"))
            self.graph.replay()
            torch.cuda.synchronize(dev)
            if not (torch.equal(self.beat, want[0]) and torch.equal(self.down, want[1])):
                raise RuntimeError("captured forward does not reproduce the plain launches (empty capture?)")

    def _launch(self):
        with torch.cuda.device(self.eng.device):
            _lib.check(_lib.lib().bt_forward_stages(self.eng._h, _lib.stream_ptr(self.eng.device), self.prec, 0, 2, self.x.data_ptr(),
                                                    self.B, self.T, self.ws.data_ptr(), self.ws.numel(), 0, self.beat.data_ptr(),
                                                    self.down.data_ptr()))

    def replay(self):
        self.graph.replay()
