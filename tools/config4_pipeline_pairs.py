#!/usr/bin/env python3
"""Config 4's per-GPU shard (32 tiles of 1024x1024, quality 0.5) through CodecPipeline with one and two encoder / decoder pairs, in jobs of
8 tiles, against one compress + one decompress of the whole shard.  usage: python tools/config4_pipeline_pairs.py -> JSON lines"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import progressivecodec_amd  # noqa: F401
import torch

from bench import source_hash
from progressivecodec_amd import CodecPipeline, synth

sd = synth.synthetic_state_dict()
x = torch.rand((32, 3, 1024, 1024), generator=torch.Generator().manual_seed(1000)).cuda()
mp = 32 * 1024 * 1024 / 1e6
for pairs in (1, 2):
    pipe = CodecPipeline(sd, device="cuda:0", n_pairs=pairs)

    def run(k):
        for _ in pipe.code({"x": x[i:i + k], "quality": 0.5} for i in range(0, 32, k)):
            pass
    for k in (8, 4):
        run(k)
        torch.cuda.synchronize()
        t0 = time.time()
        run(k); run(k)
        torch.cuda.synchronize()
        dt = (time.time() - t0) / 2
        free_b, total_b = torch.cuda.mem_get_info()
        print(json.dumps({"config": "Config 4 (one rank's shard)", "pairs": pairs, "tiles_per_job": k, "s": round(dt, 3), "megapixels_per_s": round(mp / dt, 1),
                          "hbm_gib_in_use": round((total_b - free_b) / 2 ** 30, 1), "source_hash": source_hash()}), flush=True)
    del pipe
    torch.cuda.empty_cache()
