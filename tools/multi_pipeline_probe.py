#!/usr/bin/env python3
"""Probe: do MORE kernels in flight than one encoder || decoder pair keeps (2.4 conv kernels) raise the chip's rate?  N CodecPipelines (each
its own encoder + decoder object, streams, decoder thread) driven from N host threads on the Config-2 batch; total MP/s over all of them.
usage: python tools/multi_pipeline_probe.py [steps_per_pipeline]"""
import json
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import progressivecodec_amd  # noqa: F401
import torch

from progressivecodec_amd import CodecPipeline
from progressivecodec_amd.synth import synthetic_state_dict

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
sd = synthetic_state_dict()
x = torch.rand(32, 3, 256, 256, generator=torch.Generator().manual_seed(1)).cuda()
pipes = [CodecPipeline(sd, device="cuda:0") for _ in range(3)]


def run(p, n):
    for _ in p.code({"x": x, "quality": 0.5} for _ in range(n)):
        pass


for n_pipes in (1, 2, 3, 1, 2):
    for p in pipes[:n_pipes]:
        run(p, 2)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    th = [threading.Thread(target=run, args=(p, steps)) for p in pipes[:n_pipes]]
    for t in th:
        t.start()
    for t in th:
        t.join()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(json.dumps({"pipelines": n_pipes, "steps_each": steps, "mp_per_s": round(n_pipes * steps * 32 * 256 * 256 / 1e6 / dt, 2)}), flush=True)
