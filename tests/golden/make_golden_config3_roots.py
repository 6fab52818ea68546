#!/usr/bin/env python3
"""Root flips of the Config-3 images (VERDICT r03 "Next round" 5b: "the Config-3 test prints the same per level").

At Kodak size every (image, level) pair of tests/golden/config3.json diverges from the reference in BASE slice 0 already -- a base slice,
the same for all 13 levels of an image (the base chain does not depend on the level, CHProg_cnn.py:729-767).  So the float-rounding flips
that separate the numeric contract from the reference on Config 3 are, per image, the differing elements of that one slice.  As
make_golden_config2_roots.py: the imported reference codes images 0 (512x768) and 3 (768x512) of harness.config3_images at quality 0 with
GaussianConditional.compress wrapped (symbols and indexes recorded), the contract oracle (oracle/codec_ref.py, back-end "cdet" -- what
the HIP path equals bit for bit) codes the same; the reference's planes of each image's first diverging slice are stored.

Run once in the build container:  python3 tests/golden/make_golden_config3_roots.py
Output (data only): tests/golden/config3_roots.npz -- image[n], slice[n], sym[n][49152] int16, idx[n][49152] uint8,
contract_sym_flips[n], contract_idx_flips[n]."""
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, REPO)

import ref_env  # noqa: E402

net = ref_env.canonical_model()
import torch  # noqa: E402

from oracle.codec_ref import RefCodec  # noqa: E402
from progressivecodec_amd.harness import config3_images  # noqa: E402
from progressivecodec_amd.synth import synthetic_state_dict  # noqa: E402

torch.set_num_threads(8)
sd = synthetic_state_dict()
net.load_state_dict(sd)
net.update(force=True)
gc = net.gaussian_conditional
calls = []
_orig = gc.compress


def _recording_compress(inputs, indexes, means=None):
    calls.append((gc.quantize(inputs, "symbols", means).reshape(inputs.shape[0], -1).numpy().copy(), indexes.reshape(inputs.shape[0], -1).numpy().copy()))
    return _orig(inputs, indexes, means)


gc.compress = _recording_compress
orc = RefCodec(sd, "cdet")
orc.update()
imgs = config3_images()
image, slc, syms, idxs, fs, fi = [], [], [], [], [], []
for i in (0, 3):
    x = imgs[i]
    calls.clear()
    t0 = time.perf_counter()
    with torch.no_grad():
        ref = net.compress(x, quality=0.0, mask_pol="point-based-std")
    t1 = time.perf_counter()
    taps = {}
    con = orc.compress(x, 0.0, taps=taps)
    print(f"image {i} {tuple(x.shape[2:])}: reference {t1 - t0:.1f} s, contract oracle {time.perf_counter() - t1:.1f} s, {len(calls)} base slices", flush=True)
    assert len(calls) == 10 and ref["strings"][1] == con["strings"][1], "hyper-latent strings must be identical (they are in config3.json)"
    s = next((k for k in range(10) if ref["strings"][0][k][0] != con["strings"][0][k][0]), None)
    if s is None:
        print(f"image {i}: base strings identical")
        continue
    rs, ri = calls[s][0][0], calls[s][1][0]
    cs, ci = taps[f"b{s}"]["sym"].reshape(1, -1).numpy()[0], taps[f"b{s}"]["idx"].reshape(1, -1).numpy()[0]
    image.append(i); slc.append(s); syms.append(rs.astype(np.int16)); idxs.append(ri.astype(np.uint8))
    fs.append(int((rs != cs).sum())); fi.append(int((ri != ci).sum()))
    print(f"image {i}: first diverging slice {s}, contract root flips: {fs[-1]} symbols, {fi[-1]} indexes of {rs.size}", flush=True)
np.savez_compressed(os.path.join(HERE, "config3_roots.npz"), image=np.array(image, np.int32), slice=np.array(slc, np.int32), sym=np.stack(syms),
                    idx=np.stack(idxs), contract_sym_flips=np.array(fs, np.int32), contract_idx_flips=np.array(fi, np.int32))
print("done")
