#!/usr/bin/env python3
"""Golden fixtures for compress()/decompress() with a caller-supplied importance map (`cust_map`,
/root/reference/src/compress/models/CHProg_cnn.py:686,721-722,823 / :849-851,964 -> layers/masking.py:171-194), produced by the
REAL reference imported read-only through tests/golden/ref_env.py, on the build-owned synthetic weights and seeded inputs.

Run once in the build container:   python3 tests/golden/make_golden_custmap.py
Output (committed, data only): cust_map.json -- per case sha256 + length of every byte string, mask popcounts, bpp, PSNR, x_hat hash.
The map itself is regenerated from its seed: torch.rand(B, 320, H/16, W/16, generator=manual_seed(seed + 1000)).
"""
import hashlib
import json
import math
import os
import sys

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
sha = lambda b: hashlib.sha256(b).hexdigest()


def inputs(B, H, W, seed, kind="rand"):
    g = torch.Generator().manual_seed(seed)
    if kind == "rand":
        return torch.rand(B, 3, H, W, generator=g)
    lo = torch.rand(B, 3, (H + 7) // 8, (W + 7) // 8, generator=g)
    return F.interpolate(lo, size=(H, W), mode="bilinear", align_corners=False).clamp(0, 1)


CASES = [("b2_64", 2, 64, 64, 11, "rand", [0.5, 10], "point-based-std"), ("b1_128", 1, 128, 128, 12, "smooth", [2], "two-levels")]
out = []
for name, B, H, W, seed, kind, quals, pol in CASES:
    x = inputs(B, H, W, seed, kind)
    cm = torch.rand(B, 320, H // 16, W // 16, generator=torch.Generator().manual_seed(seed + 1000))
    for q in quals:
        with torch.no_grad():
            o = net.compress(x, quality=q, mask_pol=pol, cust_map=cm)
            d = net.decompress(o["strings"], o["shape"], q, mask_pol=pol, cust_map=cm)
        ys, zs = o["strings"]
        x_hat = d["x_hat"].clamp(0, 1)
        mse = torch.mean((x - x_hat) ** 2).item()
        nbytes = sum(len(s) for sl in ys for s in sl) + sum(len(s) for s in zs)
        out.append(dict(case=name, B=B, H=H, W=W, seed=seed, kind=kind, quality=q, mask_pol=pol, shape=list(o["shape"]),
                        y_sha=[[sha(s) for s in sl] for sl in ys], z_sha=[sha(s) for s in zs],
                        mask_sums=[[int(m[b].sum().item()) for b in range(B)] for m in o["masks"]],
                        bpp=8.0 * nbytes / (B * H * W), psnr=-10.0 * math.log10(mse), x_hat_sha=sha(x_hat.numpy().tobytes())))
        print(name, q, pol, out[-1]["bpp"], out[-1]["psnr"], out[-1]["mask_sums"][0], flush=True)
json.dump(out, open(os.path.join(HERE, "cust_map.json"), "w"))
print("done")
