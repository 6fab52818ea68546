#!/usr/bin/env python3
"""Golden fixture of the REAL reference on BASELINE.json's Config 3 (VERDICT r02 "Next round" 1b): two of the 24 Kodak-sized
stand-in images of progressivecodec_amd.harness.config3_images -- index 0 (landscape 512x768) and index 3 (portrait 768x512) --
through the 13 levels of /root/reference/src/train.py:293, coded exactly as compress_with_ac does
(/root/reference/src/compress/training/step.py:318-365: centre padding to x64 -- a no-op at these sizes --, per level
compress -> decompress, clamp, PSNR, bpp from the byte-string lengths).  step.py itself is not importable here (wandb, torchvision,
pytorch_msssim are absent), so its loop is restated below around the reference's own model.

Run once in the build container:   python3 tests/golden/make_golden_config3.py
Output (data only): tests/golden/config3.json -- per image and level: sha256 + length of the 10 / 20 y strings and the z string,
mask sums, bpp, PSNR, sha256 of x_hat.
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

from progressivecodec_amd.harness import PR_LIST, config3_images  # noqa: E402
from progressivecodec_amd.synth import synthetic_state_dict  # noqa: E402

torch.set_num_threads(8)
net.load_state_dict(synthetic_state_dict())
net.update(force=True)
sha = lambda b: hashlib.sha256(b).hexdigest()
imgs = config3_images()
out = []
for idx in (0, 3):
    x = imgs[idx]
    h, w = x.shape[2:]
    pad, unpad = compute_padding(h, w, min_div=64)                                   # step.py:318
    xp = F.pad(x, pad, mode="constant", value=0)
    levels = []
    for p in PR_LIST:                                                                # step.py:322 (train.py:293)
        t0 = time.perf_counter()
        with torch.no_grad():
            data = net.compress(xp, quality=p, mask_pol="point-based-std")           # step.py:328
            dec = net.decompress(data["strings"], data["shape"], quality=p, mask_pol="point-based-std")   # step.py:333
        x_hat = F.pad(dec["x_hat"], unpad).clamp_(0, 1)                              # step.py:342-343
        ys, zs = data["strings"]
        nbytes = sum(len(s[0]) for s in ys) + sum(len(s) for s in zs)                # step.py:357-365
        levels.append(dict(quality=p, y_sha=[sha(s[0]) for s in ys], y_len=[len(s[0]) for s in ys], z_sha=sha(zs[0]), z_len=len(zs[0]),
                           mask_sums=[int(m.sum().item()) for m in data["masks"]],
                           bpp=8.0 * nbytes / (h * w), psnr=-10.0 * math.log10(torch.mean((x - x_hat) ** 2).item()),
                           x_hat_sha=sha(x_hat.numpy().tobytes())))
        print(f"image {idx} ({h}x{w}) q={p}: bpp {levels[-1]['bpp']:.6f} psnr {levels[-1]['psnr']:.6f}  {time.perf_counter() - t0:.1f} s", flush=True)
    out.append(dict(index=idx, seed=100 + idx, H=h, W=w, shape=list(data["shape"]), levels=levels))
    json.dump(dict(config="Config 3", pr_list=PR_LIST, threads=8, torch=torch.__version__, images=out), open(os.path.join(HERE, "config3.json"), "w"))
print("done")
