#!/usr/bin/env python3
"""HBM roofline of the mask / index / quantise stage: the fused
GaussianConditional kernels and the quantile threshold on one enhancement slice of Config 4's size
(256 images of 1024x1024 -> latent 64x64, 32 channels: 33.5 M elements per array), timed with HIP
events through the C ABI.  Algorithmic bytes per element: encoder enhancement
slice 32 B (the mask included), base slice 24 B, decoder index 8 B, dequantise 12 B, quantile 4 B.

usage: python tools/stage_bench.py [n_images]        prints one JSON line per kernel
"""
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from progressivecodec_amd._lib import check, lib

PEAK = 8.0e12   # HBM3E spec; ~6.3e12 achievable (MI355X_MICROARCH.md)


def timeit(fn, n=10):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / n


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    HW = 64 * 64
    L = lib()
    g = torch.Generator(device="cuda").manual_seed(0)
    n = B * HW * 32
    scale = 0.6 + 0.7 * torch.randn((B, HW, 32), device="cuda", generator=g)
    mu = torch.randn((B, HW, 32), device="cuda", generator=g)
    y = 2 * torch.randn((B, HW, 64), device="cuda", generator=g)
    table = torch.exp(torch.linspace(np.log(0.11), np.log(256), 64)).cuda()
    thr = torch.empty(B, device="cuda")
    sym = torch.empty((B, 32, HW), device="cuda", dtype=torch.int32)
    idx = torch.empty_like(sym)
    msk = torch.empty((B, 32, HW), device="cuda")
    yhat = torch.empty((B, HW, 32), device="cuda")
    P = lambda t: C.c_void_p(t.data_ptr())
    q = np.float32(0.95)

    def quant():
        check(L.pc_mask_quantile_threshold(P(scale), 32, B, HW, 32, q, P(thr), None))

    def enc_enh():
        check(L.pc_gc_prep_encode(P(scale), 32, P(mu), 32, C.c_void_p(y.data_ptr() + 128), 64, P(y), 64, P(thr), 1, B, HW,
                                  P(table), 64, 0.11, P(sym), P(idx), P(msk), P(yhat), 32, None))

    def enc_base():
        check(L.pc_gc_prep_encode(P(scale), 32, P(mu), 32, P(y), 64, None, 0, None, 0, B, HW,
                                  P(table), 64, 0.11, P(sym), P(idx), None, P(yhat), 32, None))

    def dec_idx():
        check(L.pc_gc_prep_decode_index(P(scale), 32, P(thr), 1, B, HW, P(table), 64, 0.11, P(idx), None, None))

    def deq():
        check(L.pc_gc_dequantize(P(sym), P(mu), 32, B, HW, P(yhat), 32, None))

    quant()
    for name, fn, bpe in (("quantile_thr_kernel", quant, 4), ("gc_prep_kernel<0> enhancement (mask+index+quantise+dequantise)", enc_enh, 32),
                          ("gc_prep_kernel<0> base", enc_base, 24), ("gc_prep_kernel<1> decoder index", dec_idx, 8),
                          ("gc_dequant_kernel", deq, 12)):
        t = timeit(fn)
        gbs = n * bpe / t / 1e9
        print(json.dumps({"kernel": name, "elements": n, "algorithmic_bytes_per_element": bpe, "ms": round(t * 1e3, 4),
                          "roofline": {"bound": "hbm", "achieved": round(gbs, 1), "peak": PEAK / 1e9, "unit": "GB/s",
                                       "frac": round(gbs * 1e9 / PEAK, 4)}}), flush=True)


if __name__ == "__main__":
    main()
