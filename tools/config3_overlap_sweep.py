#!/usr/bin/env python3
"""Config 3 (the 24-image Kodak-sized stand-in set x 13 levels) through compress_with_ac: one call per same-size group (batch_same_size)
against the CodecPipeline schedule (overlap=True) over group sizes; Config 4's shard (32 tiles of 1024^2) in one call against CodecPipeline
jobs of 8 / 16 tiles.  usage: python tools/config3_overlap_sweep.py  -> JSON lines"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import progressivecodec_amd  # noqa: F401  (hardware queues before HIP starts)
import torch

from bench import source_hash
from progressivecodec_amd import CodecPipeline, synth
from progressivecodec_amd.harness import PR_LIST, compress_with_ac, config3_images


def main():
    pipe = CodecPipeline(synth.synthetic_state_dict(), device="cuda:0")
    net = pipe.enc
    imgs = config3_images()
    mp = 24 * 512 * 768 * 13 / 1e6
    sh = source_hash()

    def run(label, **kw):
        compress_with_ac(pipe if kw.get("overlap") else net, imgs, PR_LIST, **kw)          # warm: workspaces of this shape on both objects
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(2):
            t0 = time.time()
            r = compress_with_ac(pipe if kw.get("overlap") else net, imgs, PR_LIST, **kw)
            torch.cuda.synchronize()
            best = min(best, time.time() - t0)
        print(json.dumps({"config": "Config 3", "schedule": label, "s": round(best, 3), "level_megapixels_per_s": round(mp / best, 1), "bpp_level0": r[0][0],
                          "source_hash": sh}), flush=True)
        return r
    base = run("batch_same_size (one call per group: 18 landscape, 6 portrait)", batch_same_size=True)
    for gs in (18, 12, 9, 6, 4, 3):
        r = run(f"overlap=True, group_size {gs}", overlap=True, group_size=gs)
        assert r[0] == base[0]
    pipe.queue_depth = 1
    run("overlap=True, group_size 6, queue depth 1", overlap=True, group_size=6)
    pipe.queue_depth = 2

    # Config 4: a rank's shard
    g = torch.Generator().manual_seed(1000)
    x = torch.rand((32, 3, 1024, 1024), generator=g).cuda()
    mp4 = 32 * 1024 * 1024 / 1e6

    def c4_one():
        d = net.compress(x, 0.5, "point-based-std")
        net.decompress(d["strings"], d["shape"], 0.5, "point-based-std")

    def c4_pipe(k):
        for _ in pipe.code({"x": x[i:i + k], "quality": 0.5} for i in range(0, 32, k)):
            pass
    for label, fn in (("one compress + one decompress of the 32 tiles", c4_one), ("CodecPipeline, jobs of 16 tiles", lambda: c4_pipe(16)),
                      ("CodecPipeline, jobs of 8 tiles", lambda: c4_pipe(8)), ("CodecPipeline, jobs of 4 tiles", lambda: c4_pipe(4))):
        fn()
        torch.cuda.synchronize()
        t0 = time.time()
        fn(); fn()
        torch.cuda.synchronize()
        dt = (time.time() - t0) / 2
        print(json.dumps({"config": "Config 4 (one rank's shard)", "schedule": label, "s": round(dt, 3), "megapixels_per_s": round(mp4 / dt, 1), "source_hash": sh}), flush=True)


if __name__ == "__main__":
    main()
