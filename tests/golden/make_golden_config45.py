#!/usr/bin/env python3
"""Golden fixtures of the REAL reference at the sizes of BASELINE.json's Config 4 and Config 5 (round 3: every configuration of BASELINE.json
now has a fixture made by the reference itself):

  Config 4: tiles 0 and 17 of rank 0's shard -- torch.rand(32,3,1024,1024) from seed 1000, the batch of
            tests/test_gpu_codec.py::test_config4_per_gpu_shard_32x1024x1024 -- each coded ALONE at quality 0.5 (an image codes
            identically alone and inside any batch: the reference loops over images, entropy_models.py:227, and the GPU test asserts it);
  Config 5: the 3840x2160 frame of test_config5_4k_frame_eight_levels_and_gather (seed 55), centre-padded to 3840x2176 as
            compress_with_ac does (step.py:318-319), at levels 0.5 and 10.

Run once in the build container (imports the reference through tests/golden/ref_env.py; ~10 minutes on 8 cores):
    python3 tests/golden/make_golden_config45.py
Output (data only): tests/golden/config45.json -- sha256 + length of every string, mask sums, bpp, PSNR, x_hat hash per case.
"""
import hashlib
import json
import math
import os
import sys
import time

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

torch.set_num_threads(8)
net.load_state_dict(synthetic_state_dict())
net.update(force=True)
sha = lambda b: hashlib.sha256(b).hexdigest()
out = []


def code(name, x, q, extra):
    h, w = x.shape[2:]
    pad, unpad = compute_padding(h, w, min_div=64)
    xp = F.pad(x, pad, mode="constant", value=0)
    t0 = time.perf_counter()
    with torch.no_grad():
        data = net.compress(xp, quality=q, mask_pol="point-based-std")
        dec = net.decompress(data["strings"], data["shape"], quality=q, mask_pol="point-based-std")
    x_hat = F.pad(dec["x_hat"], unpad).clamp_(0, 1)
    ys, zs = data["strings"]
    nbytes = sum(len(s[0]) for s in ys) + len(zs[0])
    out.append(dict(case=name, quality=q, H=h, W=w, shape=list(data["shape"]), y_sha=[sha(s[0]) for s in ys], y_len=[len(s[0]) for s in ys],
                    z_sha=sha(zs[0]), z_len=len(zs[0]), mask_sums=[int(m.sum().item()) for m in data["masks"]],
                    bpp=8.0 * nbytes / (h * w), psnr=-10.0 * math.log10(torch.mean((x - x_hat) ** 2).item()),
                    x_hat_sha=sha(x_hat.numpy().tobytes()), **extra))
    print(f"{name} q={q}: bpp {out[-1]['bpp']:.6f} psnr {out[-1]['psnr']:.6f}  {time.perf_counter() - t0:.0f} s", flush=True)
    json.dump(dict(threads=8, torch=torch.__version__, cases=out), open(os.path.join(HERE, "config45.json"), "w"))


xb = torch.rand(32, 3, 1024, 1024, generator=torch.Generator().manual_seed(1000))
for i in (0, 17):
    code(f"config4_tile{i}", xb[i:i + 1].contiguous(), 0.5, dict(config="Config 4", seed=1000, index=i))
del xb
g = torch.Generator().manual_seed(55)
lo_res = torch.rand(1, 3, 270, 480, generator=g)
x = (F.interpolate(lo_res, size=(2160, 3840), mode="bilinear", align_corners=False) + 0.03 * torch.randn(1, 3, 2160, 3840, generator=g)).clamp(0, 1)
for q in (0.5, 10):
    code("config5_frame", x, q, dict(config="Config 5", seed=55))
print("done")
