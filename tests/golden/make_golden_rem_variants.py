#!/usr/bin/env python3
"""Golden fixtures for the REM variants (VERDICT r02 "What's missing" 1): PostRateProcessedNetwork with mu_std=True (2N-channel
LatentRateReduction, the mean refined too: /root/reference/src/compress/models/CHProgREM.py:15-70, 397-416), dimension="middle" (two
ResidualBlocks per sub-net, :23-43) and escalation=True (extract_chekpoint_representation_from_images chains the check levels through
checkpoint_rep, :335-373, consumed at :773 / :989) -- produced by the REAL reference imported read-only through tests/golden/ref_env.py on
the build-owned synthetic weights (progressivecodec_amd.synth) and seeded inputs.

Run once in the build container:   python3 tests/golden/make_golden_rem_variants.py
Output (committed, data only): rem_variants.json -- per case sha256 + length of every byte string, mask popcounts, bpp, PSNR, x_hat hash,
a subsample of the refined mu / scale of slice 3 and of y_hat.
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
LEVELS = [0.01, 0.25, 1.75]
out = []
for name, mu_std, dim, esc, B, H, W, seed, kind, quals in (
        ("rem_mustd_b2_64", True, "big", False, 2, 64, 64, 11, "rand", [0.1, 2]),
        ("rem_middle_b2_64", False, "middle", False, 2, 64, 64, 11, "rand", [0.5]),
        ("rem_mustd_middle_b1_64x128", True, "middle", False, 1, 64, 128, 17, "smooth", [1.0]),
        ("rem_escalation_b1_64", False, "big", True, 1, 64, 64, 21, "rand", [1.0, 5.0])):
    rem = PostRateProcessedNetwork(base, check_levels=LEVELS, mu_std=mu_std, dimension=dim, escalation=esc).eval()
    post = synthetic_post_state_dict(3, dim, mu_std=mu_std)
    assert list(rem.post_latent.state_dict().keys()) == list(post.keys()), "post_latent layout differs from arch.rem_param_spec"
    for k, v in rem.post_latent.state_dict().items():
        assert tuple(v.shape) == tuple(post[k].shape), (k, tuple(v.shape), tuple(post[k].shape))
    rem.post_latent.load_state_dict(post)
    cap = {}
    orig = rem.apply_latent_enhancement

    def spy(current_index, quality, quality_bar, y_b_hat, mu_scale_base, mu_scale_enh, mu, scale, *a, _orig=orig, _cap=cap, **kw):
        m, s = _orig(current_index, quality, quality_bar, y_b_hat, mu_scale_base, mu_scale_enh, mu, scale, *a, **kw)
        _cap.setdefault(current_index, []).append((m.detach().clone(), s.detach().clone()))
        return m, s

    rem.apply_latent_enhancement = spy
    x = inputs(B, H, W, seed, kind)
    for q in quals:
        rep, rep_q = None, None
        with torch.no_grad():
            if esc:
                # the representation the escalation mode hands to the coder of quality q: that of the check level below it (:335-373)
                rep_q = LEVELS[1] if q <= LEVELS[2] else LEVELS[2]
                rep = rem.extract_chekpoint_representation_from_images(x, rep_q)
            cap.clear()
            o = rem.compress(x, quality=q, mask_pol="point-based-std", checkpoint_rep=rep)
            d = rem.decompress(o["strings"], o["shape"], q, mask_pol="point-based-std", checkpoint_rep=rep)
        ys, zs = o["strings"]
        x_hat = d["x_hat"].clamp(0, 1)
        nbytes = sum(len(s) for sl in ys for s in sl) + sum(len(s) for s in zs)
        out.append(dict(case=name, mu_std=mu_std, dimension=dim, escalation=esc, checkpoint_quality=rep_q, check_levels=LEVELS,
                        B=B, H=H, W=W, seed=seed, kind=kind, quality=q, shape=list(o["shape"]),
                        y_sha=[[sha(s) for s in sl] for sl in ys], z_sha=[sha(s) for s in zs],
                        mask_sums=[[int(m[b].sum().item()) for b in range(B)] for m in o["masks"]],
                        bpp=8.0 * nbytes / (B * H * W), psnr=-10.0 * math.log10(torch.mean((x - x_hat) ** 2).item()),
                        x_hat_sha=sha(x_hat.numpy().tobytes()),
                        mu3_sub=cap[3][0][0].flatten()[::37].tolist(), scale3_sub=cap[3][0][1].flatten()[::37].tolist(),
                        y_hat_sub=o["y_hat"].flatten()[::997].tolist(),
                        rep_sub=(rep.flatten()[::997].tolist() if rep is not None else None)))
        print(name, q, out[-1]["bpp"], out[-1]["psnr"], out[-1]["mask_sums"][0], flush=True)
json.dump(out, open(os.path.join(HERE, "rem_variants.json"), "w"))
print("done")
