#!/usr/bin/env python3
"""Root flips of the Config-2 batch (VERDICT r03 "Next round" 5b): WHERE the contract's float rounding first departs from the reference.

The reference (imported from /root/reference through tests/golden/ref_env.py) codes bench.py's Config-2 batch -- torch.rand(32,3,256,256)
from seed 1, quality 0.5 -- with GaussianConditional.compress wrapped so that the symbols and CDF indexes it codes are recorded
(entropy_models.py:203-238: symbols = quantize(inputs, "symbols", means)).  The numeric-contract oracle (oracle/codec_ref.py, back-end
"cdet": bit-identical to the HIP path, tests/test_gpu_codec.py) codes the same batch.  For every image whose strings differ, the
reference's symbols and indexes of the image's FIRST diverging slice are stored: a decoder of another rounding stays in step with the
reference up to that slice, so the elements that differ inside it are the float-rounding flips themselves ("root flips"); everything later
differs because its context does.

Run once in the build container:  python3 tests/golden/make_golden_config2_roots.py
Output (data only): tests/golden/config2_roots.npz
  image[n], slice[n] (-1: the hyper-latent string already differs; then no planes are stored), sym[n][8192] int16, idx[n][8192] uint8,
  contract_sym_flips[n], contract_idx_flips[n] (what the contract oracle differs by -- the GPU must reproduce these counts),
  y_sha_check: sha256 of the reference's slice-0 / image-0 string (ties the file to tests/golden/config2.json).
"""
import hashlib
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
from progressivecodec_amd.synth import synthetic_state_dict  # noqa: E402

B, S, Q, SEED = 32, 256, 0.5, 1
torch.set_num_threads(8)
sd = synthetic_state_dict()
net.load_state_dict(sd)
net.update(force=True)
x = torch.rand(B, 3, S, S, generator=torch.Generator().manual_seed(SEED))

gc = net.gaussian_conditional
calls = []
_orig = gc.compress


def _recording_compress(inputs, indexes, means=None):
    calls.append((gc.quantize(inputs, "symbols", means).reshape(B, -1).numpy().copy(), indexes.reshape(B, -1).numpy().copy()))
    return _orig(inputs, indexes, means)


gc.compress = _recording_compress
t0 = time.perf_counter()
with torch.no_grad():
    ref = net.compress(x, quality=Q, mask_pol="point-based-std")
print(f"reference compress: {time.perf_counter() - t0:.1f} s, {len(calls)} GaussianConditional.compress calls", flush=True)
assert len(calls) == 20
t0 = time.perf_counter()
orc = RefCodec(sd, "cdet")
orc.update()
taps = {}
con = orc.compress(x, Q, taps=taps)
print(f"contract oracle compress: {time.perf_counter() - t0:.1f} s", flush=True)

ry, rz = ref["strings"]
cy, cz = con["strings"]
image, slc, syms, idxs, fs, fi = [], [], [], [], [], []
for b in range(B):
    if rz[b] != cz[b]:
        image.append(b); slc.append(-1); syms.append(np.zeros(32 * 256, np.int16)); idxs.append(np.zeros(32 * 256, np.uint8)); fs.append(-1); fi.append(-1)
        continue
    s = next((k for k in range(20) if ry[k][b] != cy[k][b]), None)
    if s is None:
        continue
    rs, ri = calls[s][0][b], calls[s][1][b]
    t = taps[("b%d" % s) if s < 10 else ("e%d" % (s - 10))]
    cs, ci = t["sym"].reshape(B, -1).numpy()[b], t["idx"].reshape(B, -1).numpy()[b]
    assert np.abs(rs).max() < 32768 and ri.max() < 256
    image.append(b); slc.append(s); syms.append(rs.astype(np.int16)); idxs.append(ri.astype(np.uint8))
    fs.append(int((rs != cs).sum())); fi.append(int((ri != ci).sum()))
    print(f"image {b}: first diverging slice {s}, contract root flips: {fs[-1]} symbols, {fi[-1]} indexes", flush=True)
np.savez_compressed(os.path.join(HERE, "config2_roots.npz"), image=np.array(image, np.int32), slice=np.array(slc, np.int32),
                    sym=np.stack(syms), idx=np.stack(idxs), contract_sym_flips=np.array(fs, np.int32), contract_idx_flips=np.array(fi, np.int32),
                    y_sha_check=np.frombuffer(hashlib.sha256(ry[0][0]).hexdigest().encode(), np.uint8))
print(f"{len(image)} of {B} images diverge; root flips in total: {sum(v for v in fs if v > 0)} symbols, {sum(v for v in fi if v > 0)} indexes")
