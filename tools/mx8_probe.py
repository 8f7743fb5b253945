#!/usr/bin/env python
"""BASELINE config 5 as BASELINE.json writes it, at operator level and report-only: how fast is a GEMM with both
operands in OCP MX e4m3 (csrc/gemm_mx8.hip, v_mfma_scale_f32_32x32x64_f8f6f4) at the main layers' shapes, next to the fp16 GEMM the
half path runs there (csrc/gemm3.hip) and the hi + lo GEMM of the default path?  Each kernel looped alone, time from stream events,
joules from the SMU's accumulator (tools/smi.py).  Development tool (GPU box).

    python tools/mx8_probe.py [chunks=33]        -> gpurun_out/mx8_probe.json + a table on stdout
"""
import ctypes as C
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from beat_this_amd import _lib as L  # noqa: E402
from tools.smi import EnergyMeter  # noqa: E402

SHAPES = [("qkv", 512, 1536), ("out", 512, 512), ("ff1", 512, 2048), ("ff2", 2048, 512)]


def loop(fn, dev, seconds=1.0, energy=True):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 20
    a.record()
    for _ in range(n):
        fn()
    b.record()
    b.synchronize()
    per = a.elapsed_time(b) / n
    n = max(20, int(seconds * 1e3 / per))
    meter = EnergyMeter(dev) if energy else None
    if meter:
        meter.start()
    a.record()
    for _ in range(n):
        fn()
    b.record()
    b.synchronize()
    j = meter.stop() if meter else None
    return a.elapsed_time(b) / n * 1e3, (j[0] / n if j else None), (j[0] / j[1] if j else None)


def measure(chunks=33, seconds=1.0, dev=None, energy=True, with_hl32=True):
    """the four GEMMs of one main layer at M = 1500 x chunks rows: MX e4m3 (bt_gemm_mx8) next to fp16 and hi + lo (bt_gemm3, residual-type
    epilogue) -> list of rows (us, J, TFLOP/s per kernel)"""
    dev = dev or torch.device("cuda:0")
    M = 1500 * chunks
    lib = L.lib()
    st = L.stream_ptr(dev)
    rows = []
    for name, K, N in SHAPES:
        npad = (N + 255) // 256 * 256
        flop = 2.0 * M * K * N
        # ---- MX e4m3 (config 5) ------------------------------------------------------------------------------------------
        ab = torch.randint(0, 120, (M, K), dtype=torch.uint8, device=dev)          # finite e4m3 bytes
        wb = torch.randint(0, 120, (npad, K), dtype=torch.uint8, device=dev)
        asc = torch.full((M, K // 32), 127, dtype=torch.uint8, device=dev)
        wsc = torch.full((npad, K // 32), 120, dtype=torch.uint8, device=dev)
        out = torch.empty((M, N), dtype=torch.float32, device=dev)
        t8, j8, w8 = loop(lambda: L.check(lib.bt_gemm_mx8(st, ab.data_ptr(), asc.data_ptr(), wb.data_ptr(), wsc.data_ptr(), out.data_ptr(), M, N, K, N)),
                          dev, seconds, energy)
        # ---- fp16 (half path) and hi + lo (default path) on gemm3, residual-style epilogue (fp32 x + shadow) ------------------
        res = {}
        for tag, x3 in (("fp16", 0), ("hl32", 1)):
            if x3 and not with_hl32:
                continue
            eb = 2 if x3 else 1
            A = (torch.randn((M, K * eb), device=dev) * 0.1).to(torch.float16)
            W = (torch.randn((npad, K * eb), device=dev) * 0.02).to(torch.float16)
            x = torch.zeros((M, N), dtype=torch.float32, device=dev)
            xb = torch.empty((M, N * eb), dtype=torch.float16, device=dev)
            status = torch.zeros(4, dtype=torch.int32, device=dev)
            g = L.Gemm3Args()
            g.A, g.lda, g.M, g.K, g.W, g.N, g.epi = A.data_ptr(), K, M, K, W.data_ptr(), N, L.G3_RESID
            g.x, g.ldx, g.xb, g.no_resid, g.x3, g.status = x.data_ptr(), N, xb.data_ptr(), 1, x3, status.data_ptr()
            res[tag] = loop(lambda: L.check(lib.bt_gemm3(st, C.byref(g))), dev, seconds, energy)
        row = {"gemm": name, "M": M, "K": K, "N": N, "mx8_us": round(t8, 1), "mx8_J": j8 and round(j8, 4), "mx8_W": w8 and round(w8, 0),
               "mx8_TFLOPs": round(flop / t8 / 1e6, 1),
               "fp16_us": round(res["fp16"][0], 1), "fp16_J": res["fp16"][1] and round(res["fp16"][1], 4), "fp16_TFLOPs": round(flop / res["fp16"][0] / 1e6, 1),
               "fp16_over_mx8": round(res["fp16"][0] / t8, 3)}
        if "hl32" in res:
            row.update(hl32_us=round(res["hl32"][0], 1), hl32_J=res["hl32"][1] and round(res["hl32"][1], 4), hl32_TFLOPs=round(flop / res["hl32"][0] / 1e6, 1))
        rows.append(row)
    return rows


if __name__ == "__main__":
    chunks = int(sys.argv[1]) if len(sys.argv) > 1 else 33
    rows = measure(chunks)
    for row in rows:
        print(json.dumps(row), flush=True)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(rows, open(os.path.join(ROOT, "gpurun_out", "mx8_probe.json"), "w"), indent=1)
    t16 = sum(r["fp16_us"] for r in rows)
    t8 = sum(r["mx8_us"] for r in rows)
    print(f"one main layer's four GEMMs at M = {1500 * chunks}: fp16 {t16:.0f} us, MX e4m3 {t8:.0f} us -> {t16 / t8:.2f} x (epilogues: fp32 result + "
          f"half shadow for fp16, fp32 result only for MX e4m3 -- no quantising epilogue, no RoPE / GELU: an UPPER bound for config 5)")
