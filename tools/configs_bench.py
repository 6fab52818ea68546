#!/usr/bin/env python3
"""Encode+decode throughput on BASELINE.json's configurations beyond the headline one (which bench.py measures), on ONE GPU, each on
the per-GPU share of the configuration: Config 3 (Kodak-sized 512x768 images, all 13 levels, shared-base path), Config 4 (this
rank's shard: 32 tiles of 1024x1024, one level), Config 5 (one 3840x2160 frame, 8 levels).  Synthetic images and weights.
usage: python tools/configs_bench.py [reps]      prints one JSON line per configuration"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F

from bench import source_hash
from progressivecodec_amd import ChannelProgresssiveWACNN, synth
from progressivecodec_amd.harness import PR_LIST, compute_padding


SPLIT = {}


def timed(fn, reps):
    fn()
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.time() - t0) / reps


def split(name, enc, dec, reps):
    """encode and decode timed apart (a synchronisation between them: what the joint figure of `timed` does not pay)"""
    out = enc()
    dec(out)
    torch.cuda.synchronize()
    te = td = 0.0
    for _ in range(reps):
        t0 = time.time(); out = enc(); torch.cuda.synchronize(); t1 = time.time()
        dec(out); torch.cuda.synchronize(); t2 = time.time()
        te += t1 - t0; td += t2 - t1
    SPLIT[name] = {"encode_s": round(te / reps, 4), "decode_s": round(td / reps, 4)}


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    net = ChannelProgresssiveWACNN(device="cuda:0")
    net.load_state_dict(synth.synthetic_state_dict())
    net.update()
    g = torch.Generator().manual_seed(3)
    pol = "point-based-std"
    sh = source_hash()

    def report(name, workload, mp_levels, dt, extra):
        print(json.dumps({"config": name, "workload": workload, "s": round(dt, 4), "level_megapixels_per_s": round(mp_levels / dt, 2),
                          "n_gpus": 1, "data": "synthetic", "source_hash": sh, **extra}), flush=True)

    # Config 3
    x = torch.rand((8, 3, 512, 768), generator=g).cuda()
    levels = list(PR_LIST)

    def c3():
        ds = net.compress_levels(x, levels, pol)
        net.decompress_levels([d["strings"] for d in ds], ds[0]["shape"], levels, pol)
    split("c3", lambda: net.compress_levels(x, levels, pol), lambda ds: net.decompress_levels([d["strings"] for d in ds], ds[0]["shape"], levels, pol), reps)
    report("Config 3", "8 images of 512x768, 13 levels, compress_levels + decompress_levels", 8 * 512 * 768 * 13 / 1e6, timed(c3, reps), SPLIT["c3"])
    del x

    # Config 4 (per-GPU shard)
    x = torch.rand((32, 3, 1024, 1024), generator=g).cuda()

    def c4():
        d = net.compress(x, 0.5, pol)
        net.decompress(d["strings"], d["shape"], 0.5, pol)
    free0 = torch.cuda.mem_get_info()[0]
    dt = timed(c4, reps)
    report("Config 4 (one rank's shard)", "32 tiles of 1024x1024, quality 0.5, compress + decompress", 32 * 1024 * 1024 / 1e6, dt,
           {"hbm_gib_in_use": round((torch.cuda.mem_get_info()[1] - torch.cuda.mem_get_info()[0]) / 2 ** 30, 1)})
    del x

    # Config 5 (one frame per rank)
    lo = torch.rand(1, 3, 270, 480, generator=g)
    f = (F.interpolate(lo, size=(2160, 3840), mode="bilinear", align_corners=False) + 0.03 * torch.randn(1, 3, 2160, 3840, generator=g)).clamp(0, 1)
    pad, _ = compute_padding(2160, 3840)
    xp = F.pad(f, pad).cuda()
    lv8 = [0.05, 0.25, 0.5, 1, 2, 3, 5, 10]

    def c5():
        ds = net.compress_levels(xp, lv8, pol)
        net.decompress_levels([d["strings"] for d in ds], ds[0]["shape"], lv8, pol)
    split("c5", lambda: net.compress_levels(xp, lv8, pol), lambda ds: net.decompress_levels([d["strings"] for d in ds], ds[0]["shape"], lv8, pol), reps)
    report("Config 5 (one frame per rank)", "one 3840x2160 frame (padded to 3840x2176), 8 levels, compress_levels + decompress_levels",
           2160 * 3840 * 8 / 1e6, timed(c5, reps), SPLIT["c5"])


if __name__ == "__main__":
    main()
