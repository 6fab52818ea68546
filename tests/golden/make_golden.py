#!/usr/bin/env python3
"""Generate the golden fixtures in tests/golden/ by running the REAL reference
(/root/reference, imported read-only through tests/golden/ref_env.py) on the
build-owned synthetic weights and seeded inputs.

Run once in the build container:   python3 tests/golden/make_golden.py
Outputs (committed; data only -- inputs are regenerated from seeds, outputs are stored):
  tables.npz        GaussianConditional / EntropyBottleneck CDF tables after net.update(force=True),
                    scale_table, a few float PMF rows with their quantised CDFs (ops.cpp KATs)
  kat_rans.json     rANS known-answer vectors from the reference's own `ans` module
  quantile.npz      torch.quantile cases (the ATen op behind layers/masking.py:218)
  e2e.json          per case (B,H,W,seed,quality): sha256 + length of every byte string,
                    mask popcounts, PSNR / bpp computed as training/step.py:349-365 does
  stages_*.npz      integer-stage inputs/outputs of slices 0 and 10 (scale, mu, y -> mask, index, symbols)
  layers.npz        strided subsamples of float layer outputs (tolerance tests for the cdet oracle / HIP path)
"""
import hashlib
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
from compressai import ans, _CXX  # noqa: E402  (built from the reference's vendored sources)

from progressivecodec_amd.synth import synthetic_state_dict  # noqa: E402

torch.set_num_threads(8)
sd = synthetic_state_dict()
net.load_state_dict(sd)
net.update(force=True)
gc, eb = net.gaussian_conditional, net.entropy_bottleneck
SUB = 61  # subsample stride for float layer goldens


def sha(b):
    return hashlib.sha256(b).hexdigest()


def inputs(B, H, W, seed, kind="rand"):
    g = torch.Generator().manual_seed(seed)
    if kind == "rand":
        return torch.rand(B, 3, H, W, generator=g)
    lo = torch.rand(B, 3, (H + 7) // 8, (W + 7) // 8, generator=g)
    return F.interpolate(lo, size=(H, W), mode="bilinear", align_corners=False).clamp(0, 1)


# ------------------------------------------------------------------ tables + pmf KATs
mult = -__import__("scipy.stats").stats.norm.ppf(1e-9 / 2)
pmf_rows = {}
with torch.no_grad():
    pc = torch.ceil(gc.scale_table * mult).int()
    for r in (0, 1, 7, 20, 40, 63):
        n = int(2 * pc[r] + 1)
        samples = torch.abs(torch.arange(n).int() - pc[r]).float()
        s = gc.scale_table[r].float()
        up = gc._standardized_cumulative((0.5 - samples) / s)
        lw = gc._standardized_cumulative((-0.5 - samples) / s)
        prob = torch.cat(((up - lw), 2 * lw[:1]))
        pmf_rows[f"pmf_{r}"] = prob.numpy()
        pmf_rows[f"cdf_{r}"] = np.asarray(_CXX.pmf_to_quantized_cdf(prob.tolist(), 16), np.uint32)
        assert np.array_equal(pmf_rows[f"cdf_{r}"], gc._quantized_cdf[r, : n + 2].numpy().astype(np.uint32))
pmf_rows["pmf_a"] = np.array([0.1, 0.7, 0.15, 0.05], np.float32)
pmf_rows["cdf_a"] = np.asarray(_CXX.pmf_to_quantized_cdf([0.1, 0.7, 0.15, 0.05], 16), np.uint32)
pmf_rows["pmf_b"] = np.array([1e-9, 0.999, 1e-9, 1e-3], np.float32)
pmf_rows["cdf_b"] = np.asarray(_CXX.pmf_to_quantized_cdf([1e-9, 0.999, 1e-9, 1e-3], 16), np.uint32)
np.savez_compressed(
    os.path.join(HERE, "tables.npz"),
    gc_cdf=gc._quantized_cdf.numpy(), gc_len=gc._cdf_length.numpy(), gc_off=gc._offset.numpy(),
    eb_cdf=eb._quantized_cdf.numpy(), eb_len=eb._cdf_length.numpy(), eb_off=eb._offset.numpy(),
    scale_table=gc.scale_table.numpy(), **pmf_rows)

# ------------------------------------------------------------------ rANS KATs
rng = np.random.default_rng(7)
cdfs, lens, offs = gc._quantized_cdf.tolist(), gc._cdf_length.tolist(), gc._offset.tolist()
kats = [dict(name="survey_kat1", cdfs=[[0, 8192, 57344, 61440, 65536]], sizes=[5], offsets=[-1],
             symbols=[0, 1, -1, 0, 7, -4], indexes=[0] * 6)]
kats.append(dict(name="zeros8192", cdfs=[[0, 8192, 57344, 61440, 65536]], sizes=[5], offsets=[-1],
                 symbols=[0] * 8192, indexes=[0] * 8192))
for name, n, smax, amp in (("gc_small", 257, 12, 1.0), ("gc_mixed", 4096, 40, 1.5), ("gc_bypass_heavy", 1500, 63, 30.0),
                            ("gc_len3", 3, 5, 1.0)):
    idx = rng.integers(0, smax + 1, n)
    st = gc.scale_table.numpy()[idx]
    sym = np.rint(rng.normal(0, 1, n) * st * amp).astype(np.int64)
    if name == "gc_bypass_heavy":
        sym[::97] = rng.integers(-70000, 70000, sym[::97].size)   # multi-nibble + >15-nibble-count escapes
    kats.append(dict(name=name, table="gc", symbols=sym.tolist(), indexes=idx.tolist()))
for k in kats:
    if k.get("table") == "gc":
        enc = ans.RansEncoder().encode_with_indexes(k["symbols"], k["indexes"], cdfs, lens, offs)
        dec = ans.RansDecoder().decode_with_indexes(enc, k["indexes"], cdfs, lens, offs)
    else:
        enc = ans.RansEncoder().encode_with_indexes(k["symbols"], k["indexes"], k["cdfs"], k["sizes"], k["offsets"])
        dec = ans.RansDecoder().decode_with_indexes(enc, k["indexes"], k["cdfs"], k["sizes"], k["offsets"])
    assert list(dec) == list(k["symbols"])
    k["encoded_hex"] = enc.hex()
json.dump(kats, open(os.path.join(HERE, "kat_rans.json"), "w"))

# ------------------------------------------------------------------ torch.quantile cases
qv, qq, qr = [], [], []
for n in (7, 33, 512, 2048, 8192, 49152):
    for pr in (0.05, 0.1, 0.25, 0.5, 0.6, 0.75, 1, 1.25, 2, 3, 5, 7.5, 9.99):
        v = (rng.normal(0.6, 0.7, n)).astype(np.float32)
        if n == 512:
            v = np.round(v * 4) / 4       # many ties
        q = 1.0 - pr * 0.1                # masking.py:212-213
        qv.append(v); qq.append(q); qr.append(torch.quantile(torch.from_numpy(v), q).item())
np.savez_compressed(os.path.join(HERE, "quantile.npz"), q=np.array(qq, np.float64), r=np.array(qr, np.float32),
                    n=np.array([len(v) for v in qv]), v=np.concatenate(qv))

# ------------------------------------------------------------------ stage capture helpers
cap = {}
_orig_bi, _orig_cmp, _orig_mask = gc.build_indexes, gc.compress, net.masking.forward


def _bi(scales):
    r = _orig_bi(scales)
    cap.setdefault("bi", []).append((scales.clone(), r.clone()))
    return r


def _cmp(inp, idx, means=None, flag=1):
    cap.setdefault("cmp", []).append((inp.clone(), None if means is None else means.clone(),
                                      gc.quantize(inp, "symbols", means).clone()))
    return _orig_cmp(inp, idx, means)


def _mask(scale, **kw):
    r = _orig_mask(scale, **kw)
    cap.setdefault("mask", []).append((scale.clone(), r.clone()))
    return r


gc.build_indexes, gc.compress, net.masking.forward = _bi, _cmp, _mask

layer_out = {}


def hook(name):
    def f(m, i, o):
        layer_out[name] = (i[0].detach().clone(), o.detach().clone())
    return f


HOOKS = {"g_a.0": net.g_a[0], "g_a.1": net.g_a[1], "g_a.2": net.g_a[2], "g_a.4": net.g_a[4],
         "g_a.4.conv_b.0": net.g_a[4].conv_b[0], "g_a.4.conv_a.0": net.g_a[4].conv_a[0],
         "g_a.6": net.g_a[6], "g_a.7": net.g_a[7], "g_a.8": net.g_a[8], "g_a.8.conv_b.0": net.g_a[8].conv_b[0], "g_a": net.g_a, "h_a": net.h_a,
         "h_mean_s.0": net.h_mean_s[0], "h_scale_s.1": net.h_scale_s[1], "h_mean_s.0.2": net.h_mean_s[0][2],
         "cc_mean_transforms.0": net.cc_mean_transforms[0], "cc_scale_transforms.7": net.cc_scale_transforms[7],
         "lrp_transforms.3": net.lrp_transforms[3], "cc_mean_transforms_prog.6": net.cc_mean_transforms_prog[6],
         "lrp_transforms_prog.9": net.lrp_transforms_prog[9],
         "g_s.1.0": net.g_s[1][0], "g_s.1.1": net.g_s[1][1], "g_s.1.2": net.g_s[1][2], "g_s.1.5": net.g_s[1][5],
         "g_s.1.8": net.g_s[1][8], "g_s.1": net.g_s[1], "g_s.0": net.g_s[0]}

# ------------------------------------------------------------------ end-to-end cases
CASES = [  # name, B, H, W, seed, kind, qualities
    ("b2_64", 2, 64, 64, 11, "rand", [0, 0.05, 0.5, 5, 10]),
    ("b1_128", 1, 128, 128, 12, "smooth", [0, 0.5, 2]),
    ("b1_64x192", 1, 64, 192, 13, "rand", [0.75]),
    ("pad_96x160", 1, 96, 160, 14, "rand", [0.5]),
    ("b1_256", 1, 256, 256, 15, "smooth", [0.5]),
]
e2e = []
from compressai.ops import compute_padding  # noqa: E402
for name, B, H, W, seed, kind, quals in CASES:
    x = inputs(B, H, W, seed, kind)
    pad, unpad = compute_padding(H, W, min_div=64)                       # training/step.py:318
    xp = F.pad(x, pad, mode="constant", value=0)
    for q in quals:
        cap.clear()
        want_layers = (name == "b2_64" and q == 0.5)
        hs = [m.register_forward_hook(hook(n)) for n, m in HOOKS.items()] if want_layers else []
        with torch.no_grad():
            out = net.compress(xp, quality=q, mask_pol="point-based-std")
            dec = net.decompress(out["strings"], out["shape"], q, mask_pol="point-based-std")
        for h in hs:
            h.remove()
        x_hat = F.pad(dec["x_hat"], unpad).clamp_(0, 1)                  # step.py:342-343
        mse = torch.mean((x - x_hat) ** 2).item()
        ys, zs = out["strings"]
        nbytes = sum(len(s) for sl in ys for s in sl) + sum(len(s) for s in zs)
        e2e.append(dict(
            case=name, B=B, H=H, W=W, seed=seed, kind=kind, quality=q, shape=list(out["shape"]),
            y_sha=[[sha(s) for s in sl] for sl in ys], y_len=[[len(s) for s in sl] for sl in ys],
            z_sha=[sha(s) for s in zs], z_len=[len(s) for s in zs],
            mask_sums=[[int(m[b].sum().item()) for b in range(B)] for m in out["masks"]],
            bpp=8.0 * nbytes / (B * H * W), psnr=-10.0 * math.log10(mse),
            x_hat_sha=sha(x_hat.numpy().tobytes()),
            x_hat_sub=x_hat.flatten()[::SUB * 7].numpy().tolist()))
        print(name, q, "bpp %.4f psnr %.4f" % (e2e[-1]["bpp"], e2e[-1]["psnr"]), flush=True)
        if name in ("b2_64", "pad_96x160") and q in (0.5, 0.05):
            st = {}
            for k, (sc, idx) in enumerate(cap["bi"]):
                if k in (0, 4, 10, 13, 19):
                    st[f"bi_scale_{k}"], st[f"bi_idx_{k}"] = sc.numpy(), idx.numpy()
            for k, (inp, mu, sym) in enumerate(cap["cmp"]):
                if k in (0, 4, 10, 13, 19):
                    st[f"q_in_{k}"], st[f"q_sym_{k}"] = inp.numpy(), sym.numpy()
                    if mu is not None:
                        st[f"q_mu_{k}"] = mu.numpy()
            for k, (sc, m) in enumerate(cap["mask"]):
                if k in (0, 3, 9):
                    st[f"m_scale_{k}"], st[f"m_mask_{k}"] = sc.numpy(), m.numpy()
            np.savez_compressed(os.path.join(HERE, f"stages_{name}_q{q}.npz"), **st)
        if want_layers:
            lay = {}
            for n, (i, o) in layer_out.items():
                lay[n + "|in_shape"] = np.array(i.shape)
                lay[n + "|out_shape"] = np.array(o.shape)
                lay[n + "|out_sub"] = o.flatten()[::SUB].numpy()
                lay[n + "|out_absmax"] = np.array(o.abs().max().item(), np.float32)
                if n in ("g_a.6", "g_a.4.conv_b.0", "g_a.8.conv_b.0", "g_s.1.2", "cc_mean_transforms.0", "g_s.1.1", "g_a.7", "h_mean_s.0.2"):
                    lay[n + "|in"] = i.numpy()        # inputs for isolated-layer tests
                    lay[n + "|out"] = o.numpy()
            np.savez_compressed(os.path.join(HERE, "layers.npz"), **lay)
json.dump(e2e, open(os.path.join(HERE, "e2e.json"), "w"))
print("done")
