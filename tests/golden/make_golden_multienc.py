#!/usr/bin/env python3
"""Golden fixtures for the multiple_encoder=True topology (two analysis transforms whose outputs are concatenated,
/root/reference/src/compress/models/CHProg_cnn.py:131-144, used at :691-697 / :1011-1016) and for
forward_single_quality(force_enhanced=True) (:1006,1022,1064), produced by the REAL reference imported read-only through
tests/golden/ref_env.py on the build-owned synthetic weights (progressivecodec_amd.synth, CodecConfig(multiple_encoder=True)).

Run once in the build container:   python3 tests/golden/make_golden_multienc.py
Output (committed, data only): multienc.json -- per case sha256 + length of every byte string, mask popcounts, bpp, PSNR, x_hat hash;
"forced" entries: estimated bits and x_hat hash of forward_single_quality(quality=0, force_enhanced=True) on the canonical model.
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

ref_env.setup()
import torch  # noqa: E402
from compress.models import ChannelProgresssiveWACNN  # noqa: E402

from progressivecodec_amd.arch import CodecConfig  # noqa: E402
from progressivecodec_amd.synth import synthetic_state_dict  # noqa: E402
from tests.util import inputs  # noqa: E402

torch.set_num_threads(8)
sha = lambda b: hashlib.sha256(b).hexdigest()
out = {"multienc": [], "forced": []}

torch.manual_seed(0)
net = ChannelProgresssiveWACNN(N=192, M=640, division_dimension=[320, 640], dim_chunk=32, multiple_decoder=True, multiple_encoder=True,
                               multiple_hyperprior=True, mask_policy="two-levels", lmbda_list=[0.0055, 0.04], joiner_policy="res",
                               support_progressive_slices=5, delta_encode=True).eval()
sd = synthetic_state_dict(CodecConfig(multiple_encoder=True))
assert list(net.state_dict().keys()) == list(sd.keys()), "state_dict layout of multiple_encoder=True differs from arch.param_spec"
net.load_state_dict(sd)
net.update(force=True)
for name, B, H, W, seed, kind, quals in (("me_b2_64", 2, 64, 64, 11, "rand", [0, 0.5]), ("me_b1_64x128", 1, 64, 128, 17, "smooth", [2])):
    x = inputs(B, H, W, seed, kind)
    for q in quals:
        with torch.no_grad():
            o = net.compress(x, quality=q, mask_pol="point-based-std")
            d = net.decompress(o["strings"], o["shape"], q, mask_pol="point-based-std")
        ys, zs = o["strings"]
        x_hat = d["x_hat"].clamp(0, 1)
        nbytes = sum(len(s) for sl in ys for s in sl) + sum(len(s) for s in zs)
        out["multienc"].append(dict(case=name, B=B, H=H, W=W, seed=seed, kind=kind, quality=q, shape=list(o["shape"]),
                                    y_sha=[[sha(s) for s in sl] for sl in ys], z_sha=[sha(s) for s in zs],
                                    mask_sums=[[int(m[b].sum().item()) for b in range(B)] for m in o["masks"]],
                                    bpp=8.0 * nbytes / (B * H * W), psnr=-10.0 * math.log10(torch.mean((x - x_hat) ** 2).item()),
                                    x_hat_sha=sha(x_hat.numpy().tobytes())))
        print(name, q, out["multienc"][-1]["bpp"], out["multienc"][-1]["psnr"], flush=True)

net = ref_env.canonical_model()
net.load_state_dict(synthetic_state_dict())
net.update(force=True)
for name, B, H, W, seed, kind in (("fe_b2_64", 2, 64, 64, 11, "rand"), ("fe_b1_64x128", 1, 64, 128, 17, "smooth")):
    x = inputs(B, H, W, seed, kind)
    with torch.no_grad():
        o = net.forward_single_quality(x, 0, mask_pol="point-based-std", force_enhanced=True)
    ly, lz = o["likelihoods"]["y"], o["likelihoods"]["z"]
    out["forced"].append(dict(case=name, B=B, H=H, W=W, seed=seed, kind=kind, y_shape=list(ly.shape),
                              bits_y=float(-torch.log2(ly.double()).sum()), bits_z=float(-torch.log2(lz.double()).sum()),
                              y_sub=ly.flatten()[::53].tolist(), x_hat_sha=sha(o["x_hat"].numpy().tobytes()),
                              psnr=-10.0 * math.log10(torch.mean((x - o["x_hat"]) ** 2).item())))
    print(name, out["forced"][-1]["bits_y"], out["forced"][-1]["psnr"], flush=True)
json.dump(out, open(os.path.join(HERE, "multienc.json"), "w"))
print("done")
