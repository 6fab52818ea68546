#!/usr/bin/env python3
"""Golden fixture with the REAL reference's byte strings (not only their hashes) and its reconstruction, so that the product's
decoder can be fed bytes the reference produced (VERDICT r01 "What's weak" 1: decompress() had never seen a reference-made byte).

Run once in the build container:   python3 tests/golden/make_golden_strings.py
Outputs (data only):
  ref_strings.json   per case: every y / z byte string of net.compress() as hex, shape, per-image PSNR of the reference's x_hat
  ref_xhat.npz       the reference's x_hat (float32, un-padded and clamped as training/step.py:342-343 does), one array per case
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
from compressai.ops import compute_padding  # noqa: E402

from progressivecodec_amd.synth import synthetic_state_dict  # noqa: E402
from tests.util import inputs  # noqa: E402

torch.set_num_threads(8)
net.load_state_dict(synthetic_state_dict())
net.update(force=True)

CASES = [  # name, B, H, W, seed, kind, quality
    ("s_b2_64_q0.5", 2, 64, 64, 11, "rand", 0.5),
    ("s_b1_128_q2", 1, 128, 128, 12, "smooth", 2),
    ("s_b2_64_q0", 2, 64, 64, 11, "rand", 0),
    ("s_pad_96x160_q0.5", 1, 96, 160, 14, "rand", 0.5),
    ("s_b3_64x128_q5", 3, 64, 128, 16, "smooth", 5),
]
meta, xh = [], {}
for name, B, H, W, seed, kind, q in CASES:
    x = inputs(B, H, W, seed, kind)
    pad, unpad = compute_padding(H, W, min_div=64)                           # training/step.py:318
    xp = F.pad(x, pad, mode="constant", value=0)
    with torch.no_grad():
        out = net.compress(xp, quality=q, mask_pol="point-based-std")
        dec = net.decompress(out["strings"], out["shape"], q, mask_pol="point-based-std")
    x_hat = F.pad(dec["x_hat"], unpad).clamp_(0, 1)                          # step.py:342-343
    ys, zs = out["strings"]
    meta.append(dict(case=name, B=B, H=H, W=W, seed=seed, kind=kind, quality=q, shape=list(out["shape"]),
                     y_hex=[[s.hex() for s in sl] for sl in ys], z_hex=[s.hex() for s in zs],
                     psnr_per_image=[-10.0 * math.log10(torch.mean((x[b] - x_hat[b]) ** 2).item()) for b in range(B)],
                     psnr=-10.0 * math.log10(torch.mean((x - x_hat) ** 2).item())))
    xh[name] = x_hat.numpy()
    print(name, "bytes", sum(len(s) for sl in ys for s in sl) + sum(len(s) for s in zs), "psnr %.5f" % meta[-1]["psnr"], flush=True)
json.dump(meta, open(os.path.join(HERE, "ref_strings.json"), "w"))
np.savez_compressed(os.path.join(HERE, "ref_xhat.npz"), **xh)
print("done")
