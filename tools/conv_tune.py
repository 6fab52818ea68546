#!/usr/bin/env python3
"""Micro-benchmark of the MFMA conv kernel through the C ABI (pc_conv2d_nhwc) on representative layer shapes.
usage: python tools/conv_tune.py [shape names ...] [plain-kernel tile_cfg ...]
env (pc_conv.hip): PC_CONV_BK=32|64, PC_CONV_S=2|3|4 pick the LDS-DMA kernel's K-chunk / stage count; PC_CONV_DBG bits: 1 skip the
MFMAs, 2 skip the DMA issue, 4 all DMAs read the zero page, 64 in-kernel stamps (printed below), 256 print occupancy;
PC_TUNE_ITERS launches per timing (default 20)."""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from progressivecodec_amd._lib import check, lib

SHAPES = [  # name, B, H, W, Cin, Cout, k, stride
    ("stack_L1", 32, 16, 16, 512, 224, 3, 1),
    ("stack_L2", 32, 16, 16, 224, 176, 3, 1),
    ("stack_L3", 32, 16, 16, 176, 128, 3, 1),
    ("stack_L4", 32, 16, 16, 128, 64, 3, 1),
    ("stack_L5", 32, 16, 16, 64, 32, 3, 1),
    # the codec launches cc_mean || cc_scale as one grouped launch: same block count as a batch of 64
    ("stackg_L1", 64, 16, 16, 512, 224, 3, 1),
    ("stackg_L2", 64, 16, 16, 224, 176, 3, 1),
    ("stackg_L3", 64, 16, 16, 176, 128, 3, 1),
    ("stackg_L4", 64, 16, 16, 128, 64, 3, 1),
    ("stackg_L5", 64, 16, 16, 64, 32, 3, 1),
    ("ga_conv2", 32, 128, 128, 192, 192, 5, 2),
    ("ru_3x3", 32, 64, 64, 96, 96, 3, 1),
    ("ru_1x1", 32, 64, 64, 96, 192, 1, 1),
    ("gdn_like", 32, 128, 128, 192, 192, 1, 1),
    ("ga_conv4", 32, 32, 32, 192, 640, 5, 2),
    ("ga_conv3", 32, 64, 64, 192, 192, 5, 2),
    ("ru_1x1a", 32, 64, 64, 192, 96, 1, 1),
    ("qkv", 32, 64, 64, 192, 576, 1, 1),
    ("wam16_3x3", 32, 16, 16, 320, 320, 3, 1),
    ("wam16_1x1", 32, 16, 16, 640, 320, 1, 1),
    ("ha_0", 32, 16, 16, 640, 320, 3, 1),
    ("gs_d6", 32, 64, 64, 192, 192, 5, -1),      # stride -1: ConvTranspose2d(5, s2) in four phases
    ("gs_d3", 32, 32, 32, 192, 192, 5, -1),
    ("hs_8x8", 32, 8, 8, 224, 256, 3, 1),        # hyper-synthesis layers: few workgroups, long K
    ("hs_4x4", 32, 4, 4, 192, 896, 3, 1),
    # Round 3 probe -- what a 3-way split of the K chain could give at most (no reduction cost): the same FLOPs as the stack layers above
    # run as three times the rows with a third of the channels each (K = 9 * Cin/3, rounded to the 16-channel granule)
    ("ks3g_L1", 192, 16, 16, 176, 224, 3, 1),
    ("ks3g_L2", 192, 16, 16, 80, 176, 3, 1),
    ("ks3g_L3", 192, 16, 16, 64, 128, 3, 1),
    ("ks3g_L4", 192, 16, 16, 48, 64, 3, 1),
    ("ks3g_L5", 192, 16, 16, 16, 32, 3, 1),
    ("ks3_L1", 96, 16, 16, 176, 224, 3, 1),
    ("ks3_L2", 96, 16, 16, 80, 176, 3, 1),
    ("ks3_L3", 96, 16, 16, 64, 128, 3, 1),
    ("ks3_L4", 96, 16, 16, 48, 64, 3, 1),
    ("ks3_L5", 96, 16, 16, 16, 32, 3, 1),
]


def main():
    names = [a for a in sys.argv[1:] if not a.isdigit()]
    cfgs = [int(a) for a in sys.argv[1:] if a.isdigit()] or [0]
    L = lib()
    rng = np.random.default_rng(0)
    for name, B, H, W, ci, co, k, s in SHAPES:
        if names and name not in names:
            continue
        x = torch.from_numpy(rng.standard_normal((B, H, W, ci)).astype(np.float32)).cuda()
        w = torch.from_numpy((rng.standard_normal((k * k, ci, co)) * 0.05).astype(np.float32)).cuda()
        b = torch.zeros(co, device="cuda")
        kind = 0
        if s < 0:
            kind, s = 1, 1
            Ho, Wo = 2 * H, 2 * W
            flops = 2.0 * B * H * W * co * k * k * ci
        else:
            Ho, Wo = (H + 2 * (k // 2) - k) // s + 1, (W + 2 * (k // 2) - k) // s + 1
            flops = 2.0 * B * Ho * Wo * co * k * k * ci
        out = torch.empty((B, Ho, Wo, co), device="cuda")
        P = lambda t: C.c_void_p(t.data_ptr())
        for cfg in cfgs:
            def run():
                check(L.pc_conv2d_nhwc(P(x), B, H, W, ci, P(w), P(b), kind, co, k, s, int(os.environ.get('PC_TUNE_ACT', '0')), cfg, P(out), None))
            try:
                run()
            except Exception as e:
                print(f"{name:10s} cfg {cfg}: {e}")
                continue
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            n = int(os.environ.get("PC_TUNE_ITERS", "20"))
            e0.record()
            for _ in range(n):
                run()
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / n
            print(f"{name:10s} M={B*Ho*Wo:7d} N={co:4d} K={k*k*ci:5d} cfg {cfg}: {us:9.1f} us  {flops/us/1e6:6.1f} TFLOP/s", flush=True)
            if int(os.environ.get("PC_CONV_DBG", "0")) & 64:
                nb = min(8192, ((B * Ho * Wo + 63) // 64) * ((co + 63) // 64))
                st = np.zeros((nb, 16), np.uint64)
                check(L.pc_debug_read_stamps(st.ctypes.data_as(C.c_void_p), nb))
                st = st.astype(np.float64)
                n = st[:, 3].mean()
                print(f"   per chunk (cycles, mean over {nb} blocks, {n:.0f} chunks): loader issue {st[:,0].mean()/n:7.0f}  dma wait {st[:,1].mean()/n:7.0f}  "
                      f"loader barrier wait {st[:,2].mean()/n:7.0f} | mfma wave0 compute {st[:,4].mean()/n:7.0f} barrier wait {st[:,5].mean()/n:7.0f} | "
                      f"wave1 compute {st[:,6].mean()/n:7.0f} barrier wait {st[:,7].mean()/n:7.0f}", flush=True)
                print(f"   loader prologue (cycles): rows/masks/offsets {st[:,12].mean():7.0f}  run table + barrier {st[:,13].mean():7.0f}  first chunk issued + landed {st[:,14].mean():7.0f}")
                print(f"   per block (cycles): prologue {st[:,8].mean():7.0f}  K loop {(st[:,4]+st[:,5]).mean():8.0f}  epilogue {st[:,9].mean():7.0f}  "
                      f"| in-kernel clock over the K loop (s_memtime / s_memrealtime x 100 MHz, median over blocks) "
                      f"{np.median(st[:,10] / np.maximum(st[:,11], 1)) * 0.1:5.2f} GHz", flush=True)


if __name__ == "__main__":
    main()
