#!/usr/bin/env python3
"""Window-attention core alone (pc_win_attention_nhwc) at the Config-2 shapes: usage  [PC_LIB=...] python tools/attention_bench.py"""
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from progressivecodec_amd._lib import LIB_PATH, check, lib

L = lib()
P = lambda t: C.c_void_p(t.data_ptr())
for (B, H, W, Cc, heads, ws, shift) in ((32, 64, 64, 192, 8, 8, 4), (32, 16, 16, 640, 8, 4, 2), (32, 16, 16, 320, 8, 4, 2)):
    T = ws * ws
    qkv = torch.randn(B, H, W, 3 * Cc, device="cuda")
    bias = torch.randn(heads, T, T, device="cuda")
    out = torch.empty(B, H, W, Cc, device="cuda")
    for _ in range(3):
        check(L.pc_win_attention_nhwc(P(qkv), P(bias), B, H, W, Cc, heads, ws, shift, P(out), None))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        check(L.pc_win_attention_nhwc(P(qkv), P(bias), B, H, W, Cc, heads, ws, shift, P(out), None))
    e1.record()
    torch.cuda.synchronize()
    print(json.dumps({"lib": os.path.basename(LIB_PATH), "shape": [B, H, W, Cc], "heads": heads, "window": ws, "us_per_call": round(1e3 * e0.elapsed_time(e1) / 20, 1),
                      "checksum": float(out.double().sum().item())}))
