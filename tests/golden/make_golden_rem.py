#!/usr/bin/env python3
"""Golden fixtures for the REM model family -- PostRateProcessedNetwork.compress()/decompress()
(/root/reference/src/compress/models/CHProgREM.py:205,375,673,896; LatentRateReduction :12) -- produced by the REAL reference imported
read-only through tests/golden/ref_env.py, on the build-owned synthetic weights (base: progressivecodec_amd.synth.synthetic_state_dict,
post_latent: synthetic_post_state_dict) and seeded inputs.

Run once in the build container:   python3 tests/golden/make_golden_rem.py
Output (committed, data only): rem.json -- per case sha256 + length of every byte string, mask popcounts, bpp, PSNR, x_hat hash, and a
subsample of the refined scale of slice 3 (float check of the LatentRateReduction CNN).
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

base = ref_env.canonical_model()
import torch  # noqa: E402
from compress.models import PostRateProcessedNetwork  # noqa: E402

from progressivecodec_amd.synth import synthetic_post_state_dict, synthetic_state_dict  # noqa: E402
from tests.util import inputs  # noqa: E402

torch.set_num_threads(8)
sha = lambda b: hashlib.sha256(b).hexdigest()
base.load_state_dict(synthetic_state_dict())
base.update(force=True)
rem = PostRateProcessedNetwork(base, check_levels=[0.01, 0.25, 1.75], mu_std=False, dimension="big").eval()
post = synthetic_post_state_dict(3, "big")
assert list(rem.post_latent.state_dict().keys()) == list(post.keys()), "post_latent layout differs from arch.rem_param_spec"
rem.post_latent.load_state_dict(post)

cap = {}
orig = rem.apply_latent_enhancement


def spy(current_index, quality, quality_bar, y_b_hat, mu_scale_base, mu_scale_enh, mu, scale, *a, **kw):
    m, s = orig(current_index, quality, quality_bar, y_b_hat, mu_scale_base, mu_scale_enh, mu, scale, *a, **kw)
    cap.setdefault(current_index, []).append(s.detach().clone())
    return m, s


rem.apply_latent_enhancement = spy
out = []
# (the last two rows, ADVICE r02: the block mask follows the caller's mask_pol, the attention mask of apply_latent_enhancement never does --
#  the call sites at CHProgREM.py:620,832,1060 do not forward it -- so with "two-levels" the refinement is still quantile-gated)
for name, B, H, W, seed, kind, quals, pol in (("rem_b2_64", 2, 64, 64, 11, "rand", [0.005, 0.1, 0.5, 2, 10], "point-based-std"),
                                              ("rem_b1_64x128", 1, 64, 128, 17, "smooth", [1.0], "point-based-std"),
                                              ("rem_b2_64_two_levels", 2, 64, 64, 11, "rand", [1.0], "two-levels"),
                                              ("rem_b1_64x128_three_levels", 1, 64, 128, 17, "smooth", [1.0], "three-levels-std")):
    x = inputs(B, H, W, seed, kind)
    for q in quals:
        cap.clear()
        with torch.no_grad():
            o = rem.compress(x, quality=q, mask_pol=pol)
            d = rem.decompress(o["strings"], o["shape"], q, mask_pol=pol)
        ys, zs = o["strings"]
        x_hat = d["x_hat"].clamp(0, 1)
        nbytes = sum(len(s) for sl in ys for s in sl) + sum(len(s) for s in zs)
        out.append(dict(case=name, B=B, H=H, W=W, seed=seed, kind=kind, quality=q, mask_pol=pol, shape=list(o["shape"]),
                        y_sha=[[sha(s) for s in sl] for sl in ys], z_sha=[sha(s) for s in zs],
                        mask_sums=[[int(m[b].sum().item()) for b in range(B)] for m in o["masks"]],
                        bpp=8.0 * nbytes / (B * H * W), psnr=-10.0 * math.log10(torch.mean((x - x_hat) ** 2).item()),
                        x_hat_sha=sha(x_hat.numpy().tobytes()),
                        scale3_sub=cap[3][0].flatten()[::37].tolist(),
                        y_hat_sub=o["y_hat"].flatten()[::997].tolist()))
        print(name, q, out[-1]["bpp"], out[-1]["psnr"], out[-1]["mask_sums"][0] if out[-1]["mask_sums"] else None, flush=True)
json.dump(out, open(os.path.join(HERE, "rem.json"), "w"))
print("done")
