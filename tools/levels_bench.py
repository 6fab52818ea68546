#!/usr/bin/env python3
"""Config 3 shape (Kodak-sized 512x768 images, all 13 progressive levels of train.py:293): one compress()+decompress() per
level, as the reference's harness does (training/step.py:322-337), against compress_levels()+decompress_levels(), which
compute g_a, h_a, z, h_s and the ten base slices once per image batch.
usage: python tools/levels_bench.py [n_images] [reps]        one JSON line"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from progressivecodec_amd import ChannelProgresssiveWACNN, synth
from progressivecodec_amd.harness import PR_LIST


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    H, W = 512, 768
    net = ChannelProgresssiveWACNN(device="cuda:0")
    net.load_state_dict(synth.synthetic_state_dict())
    net.update()
    g = torch.Generator().manual_seed(3)
    x = torch.rand((B, 3, H, W), generator=g).cuda()
    levels = list(PR_LIST)

    def per_level():
        n = 0
        for q in levels:
            d = net.compress(x, q, "point-based-std")
            net.decompress(d["strings"], d["shape"], q, "point-based-std")
            n += sum(len(s) for sl in d["strings"][0] for s in sl)
        return n

    def shared():
        ds = net.compress_levels(x, levels, "point-based-std")
        net.decompress_levels([d["strings"] for d in ds], ds[0]["shape"], levels, "point-based-std")
        return sum(len(s) for d in ds for sl in d["strings"][0] for s in sl)

    def estimate():                       # forward_single_quality per level (no entropy coding), test_epoch's path
        n = 0.0
        for q in levels:
            o = net.forward_single_quality(x, q, "point-based-std")
            n += float(-torch.log2(o["likelihoods"]["y"].double()).sum()) / 8
        return int(n)

    out = {}
    for name, fn in (("per_level_calls", per_level), ("shared_base", shared), ("forward_single_quality_estimate", estimate)):
        nbytes = fn()
        torch.cuda.synchronize()
        t0 = time.time()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        dt = (time.time() - t0) / reps
        out[name] = {"s_per_rd_curve": round(dt, 4), "level_megapixels_per_s": round(B * H * W * len(levels) / dt / 1e6, 2), "y_bytes": nbytes}
    out["speedup"] = round(out["per_level_calls"]["s_per_rd_curve"] / out["shared_base"]["s_per_rd_curve"], 3)
    out["workload"] = f"{B} images of {H}x{W}, {len(levels)} levels {levels}, encode+decode, synthetic weights"
    print(json.dumps(out))


if __name__ == "__main__":
    main()
