#!/usr/bin/env python3
"""Golden fixtures of the likelihood path: forward_single_quality of the REAL reference
(/root/reference/src/compress/models/CHProg_cnn.py:1002-1198, imported read-only through tests/golden/ref_env.py) in eval mode
on the build-owned synthetic weights and seeded inputs.

Run once in the build container:   python3 tests/golden/make_golden_forward.py
Output (committed, data only): forward.npz -- per case: sum of log2-likelihoods of y and z (the estimated bits of
training/step.py:215-267), strided subsamples of both likelihood tensors, PSNR, a subsample of x_hat.
"""
import json
import math
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, REPO)

import ref_env  # noqa: E402

net = ref_env.canonical_model()
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from progressivecodec_amd.synth import synthetic_state_dict  # noqa: E402

torch.set_num_threads(8)
net.load_state_dict(synthetic_state_dict())
net.update(force=True)
net.eval()
SUB = 53


def inputs(B, H, W, seed, kind="rand"):                 # same generator as make_golden.py / tests/util.py
    g = torch.Generator().manual_seed(seed)
    if kind == "rand":
        return torch.rand(B, 3, H, W, generator=g)
    lo = torch.rand(B, 3, (H + 7) // 8, (W + 7) // 8, generator=g)
    return F.interpolate(lo, size=(H, W), mode="bilinear", align_corners=False).clamp(0, 1)


CASES = [("b2_64", 2, 64, 64, 11, "rand", [0, 0.5, 10]), ("b1_128", 1, 128, 128, 12, "smooth", [0.05, 2]), ("b1_64x192", 1, 64, 192, 13, "rand", [0.75])]
out, meta = {}, []
for name, B, H, W, seed, kind, quals in CASES:
    x = inputs(B, H, W, seed, kind)
    for q in quals:
        with torch.no_grad():
            o = net.forward_single_quality(x, quality=q, training=False, mask_pol="point-based-std")
        ly, lz = o["likelihoods"]["y"], o["likelihoods"]["z"]
        key = f"{name}_q{q}"
        out[key + "|y_sub"] = ly.flatten()[::SUB].numpy()
        out[key + "|z_sub"] = lz.flatten()[::7].numpy()
        out[key + "|x_hat_sub"] = o["x_hat"].flatten()[::SUB * 7].numpy()
        mse = torch.mean((x - o["x_hat"]) ** 2).item()
        meta.append(dict(case=name, B=B, H=H, W=W, seed=seed, kind=kind, quality=q, y_shape=list(ly.shape), z_shape=list(lz.shape),
                         bits_y=float(-torch.log2(ly.double()).sum()), bits_z=float(-torch.log2(lz.double()).sum()),
                         psnr=-10.0 * math.log10(mse)))
        print(key, meta[-1]["bits_y"], meta[-1]["bits_z"], meta[-1]["psnr"], flush=True)
out["meta_json"] = np.frombuffer(json.dumps(meta).encode(), np.uint8)
np.savez_compressed(os.path.join(HERE, "forward.npz"), **out)
print("done")
