"""ctypes binding of libbeat_this_amd.so (C ABI declared in include/beat_this_amd.h).

The library is loaded AFTER ``import torch`` so that its HIP symbols resolve to the ROCm
runtime torch has already loaded (same SONAME libamdhip64.so.7): device pointers of torch
tensors and torch's streams are then directly usable.  There is deliberately no CPU or
PyTorch fallback: if the library is missing or a device is not a ROCm GPU, we raise.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import torch  # noqa: F401  (must precede loading the HIP library)

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(PKG_DIR, "libbeat_this_amd.so")
if os.environ.get("BT_DEV") == "1" and os.environ.get("BT_LIB_PATH"):  # development only (tools/ab.sh: A/B of two builds)
    LIB_PATH = os.environ["BT_LIB_PATH"]
SOURCES = ["gemm.hip", "gemm2.hip", "gemm3.hip", "gemm_mx8.hip", "attn.hip", "attn2.hip", "fused.hip", "fused2.hip", "qkv_front.hip", "frontend.hip", "logmel.hip",
           "tail.hip", "engine.hip"]
HEADERS = ["common.h", "chain.h", "kernels.h", "attn_x3_loop.inc", "attn_hq2_loop.inc", os.path.join("..", "..", "include", "beat_this_amd.h")]

BT_OK, BT_ERR_ARG, BT_ERR_HIP, BT_ERR_WORKSPACE = 0, -1, -2, -3
ABI_VERSION = 600   # BT_ABI_VERSION of include/beat_this_amd.h this binding was written against
PREC_F32, PREC_HALF, PREC_F32X3 = 0, 1, 3   # (2 was the withdrawn e4m3 experiment)
MAX_LAYERS = 32
PROFILE_CATEGORIES = ["stem", "qkv_gemm", "attn_flash", "out_gemm", "ff1_gemm", "ff2_gemm", "conv_gemm",
                      "linear_gemm", "head", "ff_fused", "attn_freq_fused", "layer_tail"]

GEMM_EPI_STORE, GEMM_EPI_RESID, GEMM_EPI_QKV = 0, 1, 2
GEMM_F_RMS, GEMM_F_BIAS, GEMM_F_GELU, GEMM_F_OUT_F32, GEMM_F_A_F32, GEMM_F_CONV, GEMM_F_ROWMAP = 1, 2, 4, 8, 16, 32, 64


class PairWeights(C.Structure):
    _fields_ = [("dim", C.c_int32), ("heads", C.c_int32), ("w_qkvg", C.c_void_p * 2), ("b_gates", C.c_void_p),
                ("w_out", C.c_void_p * 2), ("w_ff1", C.c_void_p * 2), ("b_ff1", C.c_void_p),
                ("w_ff2", C.c_void_p * 2), ("b_ff2", C.c_void_p),
                ("w_ff_frag", C.c_void_p * 2), ("w_qkv_frag", C.c_void_p),
                ("w_outff_frag", C.c_void_p * 2), ("w_attnff_frag", C.c_void_p * 2),
                ("w_tail_frag", C.c_void_p),
                ("w_outff_frag_x3", C.c_void_p), ("w_attnff_frag_x3", C.c_void_p),
                ("w_qkvg_x3", C.c_void_p), ("w_out_x3", C.c_void_p), ("w_ff1_x3", C.c_void_p), ("w_ff2_x3", C.c_void_p),
                ("w_qkv_frag_x3", C.c_void_p),
                ("w_qkvg_f8", C.c_void_p), ("w_out_f8", C.c_void_p), ("w_ff1_f8", C.c_void_p), ("w_ff2_f8", C.c_void_p)]


class ModelDesc(C.Structure):
    _fields_ = [("transformer_dim", C.c_int32), ("n_layers", C.c_int32), ("sum_head", C.c_int32),
                ("partial_transformers", C.c_int32),
                ("bn1_scale", C.c_void_p), ("bn1_shift", C.c_void_p), ("stem_w", C.c_void_p), ("stem_b", C.c_void_p),
                ("front", (PairWeights * 2) * 3),
                ("conv_w", (C.c_void_p * 2) * 3), ("conv_b", C.c_void_p * 3),
                ("lin_w", C.c_void_p * 2), ("lin_b", C.c_void_p),
                ("layers", PairWeights * MAX_LAYERS),
                ("head_w", C.c_void_p), ("head_b", C.c_float * 2), ("rope", C.c_void_p), ("ff_mult", C.c_int32),
                ("norm_out_g", C.c_void_p), ("head_w_raw", C.c_void_p),
                ("conv_w_x3", C.c_void_p * 3), ("lin_w_x3", C.c_void_p), ("rope_len", C.c_int32)]


class LogmelTables(C.Structure):
    _fields_ = [("window", C.c_void_p), ("twiddle", C.c_void_p), ("mel_start", C.c_void_p),
                ("mel_len", C.c_void_p), ("mel_w", C.c_void_p)]


class GemmArgs(C.Structure):
    _fields_ = [("A", C.c_void_p), ("lda", C.c_int64), ("W", C.c_void_p), ("M", C.c_int32), ("N", C.c_int32),
                ("K", C.c_int32), ("epi", C.c_int32), ("flags", C.c_int32), ("bias", C.c_void_p),
                ("out", C.c_void_p), ("ldo", C.c_int64), ("x", C.c_void_p), ("ldx", C.c_int64),
                ("conv_C2", C.c_int32), ("conv_T", C.c_int32), ("conv_F", C.c_int32), ("gates", C.c_void_p),
                ("inner", C.c_int32), ("heads", C.c_int32), ("rope", C.c_void_p), ("pdiv", C.c_int32),
                ("pmod", C.c_int32), ("map_T", C.c_int32), ("map_F", C.c_int32)]


class AttnArgs(C.Structure):
    _fields_ = [("qkv", C.c_void_p), ("ld", C.c_int64), ("gates", C.c_void_p), ("out", C.c_void_p),
                ("n_seq", C.c_int32), ("L", C.c_int32), ("heads", C.c_int32), ("inner", C.c_int32),
                ("o_div", C.c_int32), ("o_outer", C.c_int64), ("o_inner", C.c_int64), ("o_tok", C.c_int64)]


class AttnFragArgs(C.Structure):
    _fields_ = [("q", C.c_void_p), ("k", C.c_void_p), ("v", C.c_void_p), ("gates", C.c_void_p), ("out", C.c_void_p),
                ("n_seq", C.c_int32), ("L", C.c_int32), ("heads", C.c_int32), ("inner", C.c_int32),
                ("nbp", C.c_int32), ("o_div", C.c_int32), ("o_outer", C.c_int64), ("o_inner", C.c_int64),
                ("o_tok", C.c_int64), ("x3", C.c_int32), ("out_f32", C.c_int32), ("status", C.c_void_p),
                ("scratch", C.c_void_p)]


class Gemm3Args(C.Structure):
    _fields_ = [("A", C.c_void_p), ("lda", C.c_int64), ("M", C.c_int32), ("K", C.c_int32), ("W", C.c_void_p),
                ("N", C.c_int32), ("epi", C.c_int32), ("bias", C.c_void_p), ("ssq_in", C.c_void_p),
                ("ssq_parts", C.c_int32), ("out", C.c_void_p), ("ldo", C.c_int64), ("x", C.c_void_p),
                ("ldx", C.c_int64), ("xb", C.c_void_p), ("ssq_out", C.c_void_p), ("n_seq", C.c_int32),
                ("L", C.c_int32), ("nbp", C.c_int32), ("heads", C.c_int32), ("rope", C.c_void_p), ("qf", C.c_void_p),
                ("kf", C.c_void_p), ("vf", C.c_void_p), ("gates", C.c_void_p), ("b_gates", C.c_void_p),
                ("no_resid", C.c_int32), ("gelu", C.c_int32), ("conv_C2", C.c_int32),
                ("conv_T", C.c_int32), ("conv_F", C.c_int32), ("x3", C.c_int32), ("status", C.c_void_p)]


class A2BPlan(C.Structure):   # bt_a2b_plan
    _fields_ = [("n22", C.c_int64), ("n_frames", C.c_int64), ("result_words", C.c_int64), ("B", C.c_int32), ("T", C.c_int32),
                ("ws_bytes", C.c_size_t), ("off_wave22", C.c_size_t), ("off_spect", C.c_size_t), ("off_chunks", C.c_size_t),
                ("off_chunk_logits", C.c_size_t), ("off_logits", C.c_size_t), ("off_result", C.c_size_t),
                ("off_forward", C.c_size_t), ("forward_bytes", C.c_size_t)]


G3_FF1, G3_RESID, G3_QKV = 0, 1, 2
UNIT_STEM, UNIT_PARTIAL, UNIT_CONV, UNIT_LINEAR, UNIT_ATTN, UNIT_FF, UNIT_NORM, UNIT_FRONT_ATTN, UNIT_FRONT_FF = range(9)

EXPORTS = {
    "bt_last_error": (C.c_char_p, []),
    "bt_version": (C.c_int, []),
    "bt_half_is_bf16": (C.c_int, []),
    "bt_struct_sizes": (None, [C.POINTER(C.c_int32)]),
    "bt_engine_create": (C.c_int, [C.POINTER(ModelDesc), C.POINTER(C.c_void_p)]),
    "bt_engine_destroy": (None, [C.c_void_p]),
    "bt_engine_set_option": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "bt_engine_get_option": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_int)]),
    "bt_workspace_bytes": (C.c_size_t, [C.c_void_p, C.c_int, C.c_int, C.c_int]),
    "bt_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_size_t,
                             C.c_void_p, C.c_void_p]),
    "bt_forward_stages": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p,
                                    C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p]),
    "bt_forward_unit": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                  C.c_void_p, C.c_size_t]),
    "bt_split_chunks": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "bt_aggregate": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int64,
                               C.c_void_p, C.c_void_p]),
    "bt_logmel": (C.c_int, [C.c_void_p, C.POINTER(LogmelTables), C.c_void_p, C.c_int64, C.c_void_p]),
    "bt_resample": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p,
                              C.c_int64]),
    "bt_peaks": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_void_p]),
    "bt_deduplicate_peaks_host": (C.c_int, [C.c_void_p, C.c_int, C.c_double, C.c_void_p, C.POINTER(C.c_int32)]),
    "bt_postprocess_host": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_double, C.c_void_p,
                                      C.POINTER(C.c_int32), C.c_void_p, C.POINTER(C.c_int32)]),
    "bt_audio2beats_plan": (C.c_int, [C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int, C.POINTER(A2BPlan)]),
    "bt_audio2beats_enqueue": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.POINTER(LogmelTables), C.c_void_p, C.c_int64, C.c_int, C.c_int,
                                         C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_int]),
    "bt_profile_begin": (None, [C.c_void_p]),
    "bt_profile_end": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]),
    "bt_resample_batch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int64, C.c_int, C.c_int, C.c_void_p, C.c_int,
                                    C.c_void_p]),
    "bt_logmel_batch": (C.c_int, [C.c_void_p, C.POINTER(LogmelTables), C.c_void_p, C.c_int, C.c_int64, C.c_void_p]),
    "bt_split_chunks_batch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "bt_aggregate_batch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int64,
                                     C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "bt_peaks_batch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "bt_peaks_host": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.POINTER(C.c_int32)]),
    "bt_gemm": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(GemmArgs)]),
    "bt_attention": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(AttnArgs)]),
    "bt_gemm3": (C.c_int, [C.c_void_p, C.POINTER(Gemm3Args)]),
    "bt_gemm_mx8": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int64]),
    "bt_attn_frag_blocks": (C.c_int, [C.c_int]),
    "bt_attention_frag": (C.c_int, [C.c_void_p, C.POINTER(AttnFragArgs)]),
    "bt_qkv_front": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(PairWeights), C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                               C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]),
    "bt_outff_fused": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(PairWeights), C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "bt_attnff_fused": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(PairWeights), C.c_void_p, C.c_void_p, C.c_int64]),
    "bt_layer_tail": (C.c_int, [C.c_void_p, C.POINTER(PairWeights), C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p,
                                C.c_void_p]),
    "bt_ff_fused": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(PairWeights), C.c_void_p, C.c_int64]),
}


# -amdgpu-mfma-vgpr-form: MFMA results stay in VGPRs (hipcc otherwise parks the accumulators of the register-chained
# frontend kernels in AGPRs and moves them with v_accvgpr_read / _write: 2900 such moves in fused2.hip, 32 per FF step);
# frequency-direction halves 0.315 -> 0.287 ms.  (-fgpu-flush-denormals-to-zero: no effect; max-ilp scheduling: slower.)
HIPCC_FLAGS = ["-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops", "-mllvm", "-amdgpu-mfma-vgpr-form"]
# tail.hip keeps 256 accumulator registers per lane next to 128 operand registers: its accumulators MUST live in AGPRs
# (a wave has 256 architectural VGPRs + 256 AGPRs), so it is compiled without -amdgpu-mfma-vgpr-form
FLAGS_BY_SOURCE = {"tail.hip": ["-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops"]}
# compile-time switches: BT_DEV_BUILD=1 in the environment of build() compiles the development instrumentation (per-wave timing
# dumps, ablation variants read by tools/*_probe.py) into the kernels; release builds contain none of it
EXTRA_DEFINES = (["-DBT_DEV"] if os.environ.get("BT_DEV_BUILD") == "1" else []) + os.environ.get("BT_DEFINES", "").split()


def build(force: bool = False, verbose: bool = False, lib_path: str | None = None, obj_dir: str | None = None,
          defines=()) -> str:
    """Compile the HIP sources for gfx950 into libbeat_this_amd.so (in tree): one object per source file under
    beat_this_amd/build/ (rebuilt only when the file, a header or the flags changed; compiled in parallel), then one link.
    ``lib_path`` / ``obj_dir`` / ``defines``: a variant build elsewhere (tools/build_variant.py, A/B measurements)."""
    from concurrent.futures import ThreadPoolExecutor

    src_dir = os.path.join(PKG_DIR, "csrc")
    obj_dir = obj_dir or os.path.join(PKG_DIR, "build")
    LIB_PATH = lib_path or os.path.join(PKG_DIR, "libbeat_this_amd.so")
    hdrs = [os.path.join(src_dir, h) for h in HEADERS] + [os.path.abspath(__file__)]  # (this file: HIPCC_FLAGS)
    hdr_time = max(os.path.getmtime(h) for h in hdrs)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    # -packed-fp32-ops: hipcc pairs independent fp32 multiplies / adds / fmas into v_pk_*_f32, which issue at 5.5 clk per
    # instruction on gfx950 against 2 x 1.8 for the scalar forms (tools/ubench/valu_rates.hip); without the pairing the
    # forward is 3.7 % faster (frequency-direction fused halves 0.43 -> 0.32 ms), A/B on one box with tools/ab.sh
    common = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC"]
    # Objects are only comparable when they were compiled with the same switches (-DBT_HALF_BF16 objects linked next to fp16
    # ones give silently wrong results): the flags and defines of a build are hashed into a stamp file in obj_dir, and a
    # stamp that differs (or is missing) forces a full rebuild.
    import hashlib

    stamp_path = os.path.join(obj_dir, "flags.stamp")
    stamp = hashlib.sha256(repr((common[1:], HIPCC_FLAGS, sorted(FLAGS_BY_SOURCE.items()), EXTRA_DEFINES, list(defines))).encode()).hexdigest()
    try:
        if open(stamp_path).read().strip() != stamp:
            force = True
    except OSError:
        force = True
    todo, objs = [], []
    for name in SOURCES:
        src = os.path.join(src_dir, name)
        obj = os.path.join(obj_dir, name.replace(".hip", ".o"))
        objs.append(obj)
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), hdr_time):
            todo.append((src, obj))
    build.last_rebuilt = [os.path.basename(j[0]) for j in todo]   # (sources recompiled by this call: __graft_entry__ lints those)
    if not todo and os.path.exists(LIB_PATH) and all(os.path.getmtime(LIB_PATH) >= os.path.getmtime(o) for o in objs):
        return LIB_PATH
    os.makedirs(obj_dir, exist_ok=True)
    if os.path.exists(stamp_path):
        os.unlink(stamp_path)   # (a build that dies half way leaves no stamp: the next one starts over)

    def compile_one(job):
        flags = FLAGS_BY_SOURCE.get(os.path.basename(job[0]), HIPCC_FLAGS)
        cmd = [*common, *flags, *EXTRA_DEFINES, *defines, "-c", job[0], "-o", job[1]]
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, stderr=subprocess.PIPE, text=True)
        # (the host pass of hipcc does not know the device feature switch and says so once per file: not a diagnostic)
        err = "\n".join(l for l in r.stderr.splitlines() if "-packed-fp32-ops' is not a recognized feature" not in l)
        if err.strip():
            print(err, flush=True)
        if r.returncode:
            raise subprocess.CalledProcessError(r.returncode, cmd)

    with ThreadPoolExecutor(max_workers=min(len(todo), os.cpu_count() or 4) or 1) as pool:
        list(pool.map(compile_one, todo))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", LIB_PATH]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    with open(stamp_path, "w") as f:
        f.write(stamp + "\n")
    return LIB_PATH


build.last_rebuilt = []


def device_asm_command(src: str, out: str, defines=()) -> list:
    """The hipcc command line that regenerates the device ISA of ``src`` exactly as ``build`` compiles it (same flags, same
    defines) -- what tools/isa_lint.py reads."""
    flags = FLAGS_BY_SOURCE.get(os.path.basename(src), HIPCC_FLAGS)
    return [os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", *flags,
            *EXTRA_DEFINES, *defines, "-S", "--cuda-device-only", src, "-o", out]


_lib = None


def lib():
    """The loaded library; raises ImportError if it was never built (no fallback path exists)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(needs hipcc); beat_this_amd has no CPU/PyTorch fallback")
        handle = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
        handle.bt_version.restype = C.c_int
        if handle.bt_version() != ABI_VERSION:   # (a stale .so from another revision: signatures / struct layouts differ)
            raise ImportError(f"{LIB_PATH} has ABI version {handle.bt_version()}, this binding needs {ABI_VERSION}: rebuild it "
                              "(python -c 'import __graft_entry__ as g; g.build()')")
        for name, (res, args) in EXPORTS.items():
            fn = getattr(handle, name)
            fn.restype = res
            fn.argtypes = args
        _lib = handle
    return _lib


def half_dtype_name() -> str:
    """Operand type of the half-precision path this library was built with ("f16" or "bf16")."""
    return "bf16" if lib().bt_half_is_bf16() else "f16"


def half_torch_dtype():
    return torch.bfloat16 if lib().bt_half_is_bf16() else torch.float16


def check(rc: int) -> None:
    if rc == BT_OK:
        return
    msg = lib().bt_last_error().decode("utf-8", "replace")
    if rc == BT_ERR_ARG:
        raise ValueError(msg)
    raise RuntimeError(f"beat_this_amd: {msg} (code {rc})")


def require_gpu(t: torch.Tensor, what: str = "tensor") -> None:
    if not t.is_cuda:
        raise RuntimeError(
            f"beat_this_amd runs on ROCm GPUs only: {what} is on '{t.device}'. "
            "Construct the model with device='cuda' (there is no CPU path in this package).")


def stream_ptr(device) -> int:
    return torch.cuda.current_stream(device).cuda_stream


def upload(a, device) -> torch.Tensor:
    """A small host table (numpy) -> device WITHOUT blocking the host: an ordinary ``.to(device)`` of pageable memory
    synchronises the stream, i.e. waits for everything enqueued before it (a whole forward, when the table is the peak
    picker's).  The copy goes through a pinned staging block on the current stream; torch's caching host allocator keeps
    the block until the copy has run."""
    import numpy as np

    return torch.from_numpy(np.ascontiguousarray(a)).pin_memory().to(device, non_blocking=True)


def ptr(t) -> int:
    return 0 if t is None else t.data_ptr()
