"""Pins the CPU oracle against fixtures produced by the REAL reference (tests/golden/make_golden.py):
integer stages bit-exact; the ATen back-end reproduces every reference byte string; the
numeric-contract ("cdet") back-end agrees with the reference to float rounding."""
import functools
import hashlib
import json
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import liboracle as lo
from oracle.codec_ref import RefCodec, bpp_of, compute_padding, psnr_of
from tests.util import GOLD, e2e_cases, inputs, oracle_codec, synth_sd, tables_npz

sha = lambda b: hashlib.sha256(b).hexdigest()


def test_rans_known_answers():
    t = tables_npz()
    gc = lo.Tables(t["gc_cdf"], t["gc_len"], t["gc_off"])
    for k in json.load(open(os.path.join(GOLD, "kat_rans.json"))):
        tb = gc if k.get("table") == "gc" else lo.Tables(np.array(k["cdfs"]), k["sizes"], k["offsets"])
        assert lo.rans_encode(k["symbols"], k["indexes"], tb).hex() == k["encoded_hex"], k["name"]
        assert lo.rans_decode(bytes.fromhex(k["encoded_hex"]), k["indexes"], tb).tolist() == k["symbols"]


def test_pmf_to_quantized_cdf_known_answers():
    t = tables_npz()
    for key in [k[4:] for k in t if k.startswith("pmf_")]:
        assert np.array_equal(lo.pmf_to_quantized_cdf(t["pmf_" + key]), t["cdf_" + key]), key


def test_quantile_is_torch_quantile():
    z = np.load(os.path.join(GOLD, "quantile.npz"))
    off = 0
    for q, r, n in zip(z["q"], z["r"], z["n"]):
        v = z["v"][off:off + n]
        off += n
        assert lo.quantile(v, np.float32(q)) == r, (n, q)


@pytest.mark.parametrize("name", ["stages_b2_64_q0.5.npz", "stages_b2_64_q0.05.npz", "stages_pad_96x160_q0.5.npz"])
def test_integer_stages_bit_exact_given_reference_floats(name):
    """build_indexes / quantize / mask from the reference's own float tensors -> identical integers."""
    z = np.load(os.path.join(GOLD, name))
    table = tables_npz()["scale_table"]
    pr = float(name.split("_q")[1][:-4])
    n = 0
    for k in [k for k in z if k.startswith("bi_scale_")]:
        i = k.split("_")[-1]
        assert np.array_equal(lo.build_indexes(z[k], table, 0.11), z["bi_idx_" + i]); n += 1
    for k in [k for k in z if k.startswith("q_in_")]:
        i = k.split("_")[-1]
        mu = z["q_mu_" + i] if "q_mu_" + i in z else None
        assert np.array_equal(lo.quantize(z[k], mu), z["q_sym_" + i]); n += 1
    for k in [k for k in z if k.startswith("m_scale_")]:
        i = k.split("_")[-1]
        assert np.array_equal(lo.mask_point_based_std(z[k], pr), z["m_mask_" + i]); n += 1
    assert n >= 8


@pytest.mark.parametrize("idx", [0, 2, 4, 6, 9])
def test_aten_backend_reproduces_reference_bitstreams(idx):
    c = e2e_cases()[idx]
    torch.set_num_threads(8)
    x = inputs(c["B"], c["H"], c["W"], c["seed"], c["kind"])
    pad, unpad = compute_padding(c["H"], c["W"])
    orc = oracle_codec("torch")
    out = orc.compress(F.pad(x, pad), c["quality"])
    ys, zs = out["strings"]
    assert [sha(s) for s in zs] == c["z_sha"]
    assert [[sha(s) for s in sl] for sl in ys] == c["y_sha"]
    assert list(out["shape"]) == c["shape"]
    assert [[int(m[b].sum()) for b in range(c["B"])] for m in out["masks"]] == c["mask_sums"]
    x_hat = F.pad(orc.decompress(out["strings"], out["shape"], c["quality"])["x_hat"], unpad).clamp_(0, 1)
    assert sha(x_hat.numpy().tobytes()) == c["x_hat_sha"]
    assert abs(bpp_of(out["strings"], c["B"], c["H"], c["W"]) - c["bpp"]) < 1e-12
    assert abs(psnr_of(x, x_hat) - c["psnr"]) < 1e-9


@pytest.mark.parametrize("idx", [2, 9])
def test_contract_backend_agrees_with_reference_to_rounding(idx):
    c = e2e_cases()[idx]
    x = inputs(c["B"], c["H"], c["W"], c["seed"], c["kind"])
    pad, unpad = compute_padding(c["H"], c["W"])
    orc = oracle_codec("cdet")
    out = orc.compress(F.pad(x, pad), c["quality"])
    ys, zs = out["strings"]
    assert [sha(s) for s in zs] == c["z_sha"]
    n_ok = sum(sha(s) == h for sl, hl in zip(ys, c["y_sha"]) for s, h in zip(sl, hl))
    assert n_ok >= 0.9 * sum(len(sl) for sl in ys)
    x_hat = F.pad(orc.decompress(out["strings"], out["shape"], c["quality"])["x_hat"], unpad).clamp_(0, 1)
    assert abs(psnr_of(x, x_hat) - c["psnr"]) < 2e-3          # dB; isolated symbol flips from float rounding
    assert abs(bpp_of(out["strings"], c["B"], c["H"], c["W"]) - c["bpp"]) < 2e-3 * c["bpp"]


def test_float_layers_of_both_backends_match_reference_subsamples():
    """Layer outputs captured with forward hooks on the reference (b2_64, q=0.5), strided subsample."""
    z = np.load(os.path.join(GOLD, "layers.npz"))
    x = inputs(2, 64, 64, 11, "rand")
    for backend, tol in (("torch", 0.0), ("cdet", 3e-5)):
        orc = oracle_codec(backend)
        t0 = orc._c(x, "g_a.0", 2)
        t1 = orc._gdn(t0, "g_a.1")
        t2 = orc._c(t1, "g_a.2", 2)
        y = orc.g_a(x)
        zz = orc.h_a(y)
        got = {"g_a.0": t0, "g_a.1": t1, "g_a.2": t2, "g_a": y, "h_a": zz}
        for name, v in got.items():
            ref = z[name + "|out_sub"]
            sub = v.flatten()[::61].numpy()
            scale = float(z[name + "|out_absmax"])
            err = np.abs(sub - ref).max() / max(scale, 1e-6)
            assert err <= tol, (backend, name, err)
    # isolated layers with stored inputs
    orc = oracle_codec("cdet")
    cases = {
        "g_a.6": lambda i: orc._gdn(i, "g_a.6"),
        "g_s.1.2": lambda i: orc._gdn(i, "g_s.1.2", True),
        "g_a.7": lambda i: orc._c(i, "g_a.7", 2),
        "g_s.1.1": lambda i: orc.ops.deconv(i, orc.sd["g_s.1.1.weight"], orc.sd["g_s.1.1.bias"]),
        "cc_mean_transforms.0": lambda i: orc.stack5("cc_mean_transforms", 0, i),
        "h_mean_s.0.2": lambda i: F.pixel_shuffle(orc._c(i, "h_mean_s.0.2.0"), 2),
    }
    for name, fn in cases.items():
        i, o = torch.from_numpy(z[name + "|in"]), z[name + "|out"]
        got = fn(i).numpy()
        assert np.abs(got - o).max() <= 3e-5 * max(1.0, np.abs(o).max()), name
    for name, ws, sh in (("g_a.4.conv_b.0", 8, 4), ("g_a.8.conv_b.0", 4, 2)):
        p = name.rsplit(".conv_b.0", 1)[0]
        ap = {k: orc.sd[f"{p}.conv_b.0.attn.{k}"] for k in ("qkv.weight", "qkv.bias", "proj.weight", "proj.bias",
                                                            "relative_position_bias_table", "relative_position_index")}
        got = orc.ops.win_attention(torch.from_numpy(z[name + "|in"]), ap, ws, sh).numpy()
        o = z[name + "|out"]
        assert np.abs(got - o).max() <= 3e-5 * max(1.0, np.abs(o).max()), name


def test_harness_padding_and_metrics():
    """compute_padding / bpp / psnr as training/step.py:318,349-365 (centre pad to a multiple of 64)."""
    assert compute_padding(96, 160) == ((16, 16, 16, 16), (-16, -16, -16, -16))
    assert compute_padding(512, 768) == ((0, 0, 0, 0), (0, 0, 0, 0))
    assert compute_padding(100, 70)[0] == (29, 29, 14, 14)
    assert bpp_of([[[b"1234"], [b"12345678"]], [b"1234"]], 1, 4, 8) == 8 * 16 / 32


# ------------------------------------------------------------------ likelihood path
def _forward_golden():
    import json
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "forward.npz"))
    return g, json.loads(bytes(g["meta_json"]).decode())


@pytest.mark.parametrize("idx", [0, 1, 5])
def test_forward_single_quality_torch_backend_equals_reference(idx):
    """The ATen back-end of the oracle reproduces the reference's forward_single_quality: likelihood subsamples bit for bit,
    estimated bits to double rounding (fixtures: tests/golden/make_golden_forward.py)."""
    from tests.util import inputs, oracle_codec
    g, meta = _forward_golden()
    m = meta[idx]
    key = f"{m['case']}_q{m['quality']}"
    x = inputs(m["B"], m["H"], m["W"], m["seed"], m["kind"])
    out = oracle_codec("torch").forward_single_quality(x, m["quality"])
    ly, lz = out["likelihoods"]["y"], out["likelihoods"]["z"]
    assert list(ly.shape) == m["y_shape"] and list(lz.shape) == m["z_shape"]
    assert np.array_equal(ly.flatten()[::53].numpy(), g[key + "|y_sub"])
    assert np.allclose(lz.flatten()[::7].numpy(), g[key + "|z_sub"], rtol=2e-6, atol=0)
    assert abs(float(-torch.log2(ly.double()).sum()) - m["bits_y"]) <= 1e-9 * m["bits_y"]
    assert abs(float(-torch.log2(lz.double()).sum()) - m["bits_z"]) <= 1e-6 * m["bits_z"]
    assert np.array_equal(out["x_hat"].flatten()[::53 * 7].numpy(), g[key + "|x_hat_sub"])


# ------------------------------------------------------------------ cust_map masks
def _cust_cases():
    return json.load(open(os.path.join(os.path.dirname(__file__), "golden", "cust_map.json")))


def cust_map_of(c):
    return torch.rand(c["B"], 320, c["H"] // 16, c["W"] // 16, generator=torch.Generator().manual_seed(c["seed"] + 1000))


@pytest.mark.parametrize("idx", [0, 2])
def test_cust_map_torch_backend_equals_reference(idx):
    """compress()/decompress() with a caller-supplied importance map: every string, mask popcount and the x_hat hash of the
    reference (tests/golden/make_golden_custmap.py) are reproduced by the oracle's ATen back-end."""
    c = _cust_cases()[idx]
    x = inputs(c["B"], c["H"], c["W"], c["seed"], c["kind"])
    cm = cust_map_of(c)
    orc = oracle_codec("torch")
    out = orc.compress(x, c["quality"], c["mask_pol"], cust_map=cm)
    ys, zs = out["strings"]
    assert [[hashlib.sha256(s).hexdigest() for s in sl] for sl in ys] == c["y_sha"]
    assert [hashlib.sha256(s).hexdigest() for s in zs] == c["z_sha"]
    assert [[int(m[b].sum().item()) for b in range(c["B"])] for m in out["masks"]] == c["mask_sums"]
    dec = orc.decompress(out["strings"], out["shape"], c["quality"], c["mask_pol"], cust_map=cm)["x_hat"].clamp(0, 1)
    assert hashlib.sha256(dec.numpy().tobytes()).hexdigest() == c["x_hat_sha"]


# ------------------------------------------------------------------ multiple_encoder / force_enhanced
def _multienc():
    return json.load(open(os.path.join(os.path.dirname(__file__), "golden", "multienc.json")))


@functools.lru_cache(maxsize=None)
def multienc_sd():
    from progressivecodec_amd.arch import CodecConfig
    from progressivecodec_amd.synth import synthetic_state_dict
    sd = synthetic_state_dict(CodecConfig(multiple_encoder=True))
    t = tables_npz()        # cc_* / h_* / entropy-model weights are those of the canonical synthetic model: the same tables apply
    for p, k in (("gaussian_conditional", "gc"), ("entropy_bottleneck", "eb")):
        sd[p + "._quantized_cdf"] = torch.from_numpy(t[k + "_cdf"])
        sd[p + "._cdf_length"] = torch.from_numpy(t[k + "_len"])
        sd[p + "._offset"] = torch.from_numpy(t[k + "_off"])
    return sd


@pytest.mark.parametrize("idx", [0, 1, 2])
def test_multiple_encoder_torch_backend_equals_reference(idx):
    """multiple_encoder=True (CHProg_cnn.py:131-144,691-697): every byte string, mask popcount and the x_hat hash of the reference
    (tests/golden/make_golden_multienc.py) are reproduced by the oracle's ATen back-end."""
    c = _multienc()["multienc"][idx]
    torch.set_num_threads(8)
    x = inputs(c["B"], c["H"], c["W"], c["seed"], c["kind"])
    orc = RefCodec(multienc_sd(), "torch")
    out = orc.compress(x, c["quality"])
    ys, zs = out["strings"]
    assert [sha(s) for s in zs] == c["z_sha"]
    assert [[sha(s) for s in sl] for sl in ys] == c["y_sha"]
    assert [[int(m[b].sum()) for b in range(c["B"])] for m in out["masks"]] == c["mask_sums"]
    dec = orc.decompress(out["strings"], out["shape"], c["quality"])["x_hat"].clamp(0, 1)
    assert sha(dec.numpy().tobytes()) == c["x_hat_sha"]


@pytest.mark.parametrize("idx", [0, 1])
def test_force_enhanced_torch_backend_equals_reference(idx):
    """forward_single_quality(quality=0, force_enhanced=True) (CHProg_cnn.py:1006,1022,1064): likelihood subsample and x_hat hash
    of the reference reproduced by the oracle's ATen back-end."""
    c = _multienc()["forced"][idx]
    torch.set_num_threads(8)
    x = inputs(c["B"], c["H"], c["W"], c["seed"], c["kind"])
    out = oracle_codec("torch").forward_single_quality(x, 0, force_enhanced=True)
    ly = out["likelihoods"]["y"]
    assert list(ly.shape) == c["y_shape"] and ly.shape[1] == 640
    assert np.array_equal(ly.flatten()[::53].numpy(), np.asarray(c["y_sub"], np.float32))
    assert abs(float(-torch.log2(ly.double()).sum()) - c["bits_y"]) <= 1e-9 * c["bits_y"]
    assert sha(out["x_hat"].numpy().tobytes()) == c["x_hat_sha"]
    assert all(int(m.sum()) == 0 for m in out["masks"])


# ------------------------------------------------------------------ REM model family
def _rem_cases():
    return json.load(open(os.path.join(os.path.dirname(__file__), "golden", "rem.json")))


@functools.lru_cache(maxsize=None)
def rem_post_sd():
    from progressivecodec_amd.synth import synthetic_post_state_dict
    return synthetic_post_state_dict(3, "big")


def rem_oracle(backend):
    from oracle.codec_ref import RemCodec
    return RemCodec(synth_sd(), rem_post_sd(), backend)


@pytest.mark.parametrize("idx", [0, 1, 3, 4, 5, 6, 7])
def test_rem_torch_backend_equals_reference(idx):
    """PostRateProcessedNetwork.compress()/decompress() (CHProgREM.py:673,896): every byte string, mask popcount, the refined scale
    of slice 3 and the x_hat hash of the reference (tests/golden/make_golden_rem.py) reproduced by the oracle's ATen back-end --
    below the first check level (no refinement), inside each of the three refinement ranges, and at quality 10."""
    c = _rem_cases()[idx]
    torch.set_num_threads(8)
    x = inputs(c["B"], c["H"], c["W"], c["seed"], c["kind"])
    orc = rem_oracle("torch")
    taps = {}
    pol = c.get("mask_pol", "point-based-std")       # cases 6, 7: the block mask follows the policy, the attention mask does not
    out = orc.compress(x, c["quality"], pol, taps=taps)
    ys, zs = out["strings"]
    assert [sha(s) for s in zs] == c["z_sha"]
    assert [[sha(s) for s in sl] for sl in ys] == c["y_sha"]
    assert [[int(m[b].sum()) for b in range(c["B"])] for m in out["masks"]] == c["mask_sums"]
    assert np.array_equal(taps["e3"]["scale"].flatten()[::37].numpy(), np.asarray(c["scale3_sub"], np.float32))
    dec = orc.decompress(out["strings"], out["shape"], c["quality"], pol)["x_hat"].clamp(0, 1)
    assert sha(dec.numpy().tobytes()) == c["x_hat_sha"]
    assert abs(bpp_of(out["strings"], c["B"], c["H"], c["W"]) - c["bpp"]) < 1e-12


# ----------------------------------------------------------------------------- REM variants (mu_std, dimension "middle", escalation / checkpoint_rep)
def _rem_variant_cases():
    return json.load(open(os.path.join(os.path.dirname(__file__), "golden", "rem_variants.json")))


def rem_variant_oracle(c, backend):
    from oracle.codec_ref import RemCodec
    from progressivecodec_amd.synth import synthetic_post_state_dict
    post = synthetic_post_state_dict(3, c["dimension"], mu_std=c["mu_std"])
    return RemCodec(synth_sd(), post, backend, check_levels=c["check_levels"], mu_std=c["mu_std"])


def rem_variant_rep(orc, c, x):
    """the checkpoint representation the escalation mode hands to a coder of quality c["quality"] (CHProgREM.py:335-373): the y_hat of
    the check level below it, each level reading the representation of the level before"""
    if not c["escalation"]:
        return None
    lv = c["check_levels"]

    def y_hat_of(q):                                      # "y_hat" of PostRateProcessedNetwork.compress at q > 0: the merged enhancement slices
        t = {}
        orc.compress(x, q, taps=t)
        return torch.cat([t[f"e{i}"]["y_hat"] for i in range(10)], 1)
    orc.set_checkpoint_rep(None)
    rep = y_hat_of(lv[0])
    for q in lv[1:]:
        if c["checkpoint_quality"] < q:
            break
        orc.set_checkpoint_rep(rep)
        rep = y_hat_of(q)
    orc.set_checkpoint_rep(None)
    return rep


@pytest.mark.parametrize("idx", range(6))
def test_rem_variants_torch_backend_equals_reference(idx):
    """mu_std=True, dimension="middle" and escalation (checkpoint_rep) forms of PostRateProcessedNetwork (CHProgREM.py:15-70, 335-373,
    397-416, 773, 989): every byte string, mask popcount, the refined mu / scale of slice 3 and the x_hat hash of the reference
    (tests/golden/make_golden_rem_variants.py) reproduced by the oracle's ATen back-end."""
    c = _rem_variant_cases()[idx]
    torch.set_num_threads(8)
    x = inputs(c["B"], c["H"], c["W"], c["seed"], c["kind"])
    orc = rem_variant_oracle(c, "torch")
    rep = rem_variant_rep(orc, c, x)
    if rep is not None:
        assert np.array_equal(rep.flatten()[::997].numpy(), np.asarray(c["rep_sub"], np.float32))
    taps = {}
    orc.set_checkpoint_rep(rep)
    out = orc.compress(x, c["quality"], taps=taps)
    ys, zs = out["strings"]
    assert [sha(s) for s in zs] == c["z_sha"]
    assert [[sha(s) for s in sl] for sl in ys] == c["y_sha"]
    assert [[int(m[b].sum()) for b in range(c["B"])] for m in out["masks"]] == c["mask_sums"]
    assert np.array_equal(taps["e3"]["scale"].flatten()[::37].numpy(), np.asarray(c["scale3_sub"], np.float32))
    assert np.array_equal(taps["e3"]["mu"].flatten()[::37].numpy(), np.asarray(c["mu3_sub"], np.float32))
    orc.set_checkpoint_rep(rep)
    dec = orc.decompress(out["strings"], out["shape"], c["quality"])["x_hat"].clamp(0, 1)
    assert sha(dec.numpy().tobytes()) == c["x_hat_sha"]
    assert abs(bpp_of(out["strings"], c["B"], c["H"], c["W"]) - c["bpp"]) < 1e-12
