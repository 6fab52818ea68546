"""End-to-end GPU parity of compress()/decompress() through the C ABI.

* vs the CPU oracle in the numeric-contract order (backend "cdet"): every byte string, every
  mask, every symbol/index and x_hat must be IDENTICAL (bit-exact) -- same inputs, same weights.
* vs the committed goldens of the real reference (tests/golden/e2e.json): bpp and PSNR within the
  stated tolerances; byte strings equal except where float rounding flips a symbol (reported).
* size-independent properties at the bench size: encode->decode round trip consistency, batch
  invariance (an image codes identically at B=1 and inside a batch), mask popcounts.
"""
import hashlib
import math
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from oracle.codec_ref import bpp_of, compute_padding, psnr_of  # noqa: E402
from tests.util import e2e_cases, gpu_codec, inputs, oracle_codec  # noqa: E402

NORTH_STAR_PSNR_TOL_DB = 1e-4   # BASELINE.json north_star: "reconstructed pixels within 1e-4 dB PSNR" -- asserted on every flip-free case
PSNR_TOL_DB = 2e-3      # cases in which float rounding flipped a symbol against the PyTorch reference (reported one by one)
BPP_TOL = 2e-3


def sha(b):
    return hashlib.sha256(b).hexdigest()


def run_case(c):
    x = inputs(c["B"], c["H"], c["W"], c["seed"], c["kind"])
    pad, unpad = compute_padding(c["H"], c["W"])
    xp = F.pad(x, pad)
    net = gpu_codec()
    out = net.compress(xp.cuda(), c["quality"], "point-based-std")
    dec = net.decompress(out["strings"], out["shape"], c["quality"], "point-based-std")
    x_hat = F.pad(dec["x_hat"].cpu(), unpad).clamp_(0, 1)
    return x, xp, out, x_hat


@pytest.mark.parametrize("idx", [0, 2, 3, 4, 6, 8, 9])
def test_bit_exact_vs_oracle(idx):
    c = e2e_cases()[idx]
    x, xp, out, x_hat = run_case(c)
    orc = oracle_codec("cdet")
    ref = orc.compress(xp, c["quality"])
    ys, zs = out["strings"]
    rys, rzs = ref["strings"]
    assert tuple(out["shape"]) == tuple(ref["shape"])
    assert len(ys) == len(rys)
    assert zs == rzs, "z strings differ"
    for s, (a, b) in enumerate(zip(ys, rys)):
        assert a == b, f"y strings of slice {s} differ"
    for m, rm in zip(out["masks"], ref["masks"]):
        assert np.array_equal(m.cpu().numpy(), rm.numpy())
    rdec = orc.decompress(ref["strings"], ref["shape"], c["quality"])
    r_x_hat = F.pad(rdec["x_hat"], compute_padding(c["H"], c["W"])[1]).clamp_(0, 1)
    assert np.array_equal(x_hat.numpy().view(np.uint32), r_x_hat.numpy().view(np.uint32)), \
        f"x_hat max abs diff {(x_hat - r_x_hat).abs().max().item()}"


def flip_report(ys, y_sha, B):
    """per image: index of the first y slice whose string differs from the reference's (None: all identical)"""
    first = [None] * B
    for s, (sl, hl) in enumerate(zip(ys, y_sha)):
        for b in range(B):
            if first[b] is None and sha(sl[b]) != hl[b]:
                first[b] = s
    return first


@pytest.mark.parametrize("idx", range(11))
def test_vs_reference_goldens(idx):
    """Against the REAL reference's fixtures (tests/golden/e2e.json).  The hyper-latent strings, mask popcounts and shapes must
    be identical.  An image none of whose y strings differs ("flip-free") must meet the north-star tolerance: identical byte
    count, PSNR within 1e-4 dB.  An image in which float rounding (oneDNN's summation order vs the contract's) flipped a symbol
    differs from that slice on; those are listed with their first diverging slice and held to 2e-3 dB / 2e-3 relative bpp."""
    c = e2e_cases()[idx]
    x, xp, out, x_hat = run_case(c)
    ys, zs = out["strings"]
    assert list(out["shape"]) == c["shape"]
    assert [sha(s) for s in zs] == c["z_sha"], "hyper-latent strings must match the reference exactly"
    first = flip_report(ys, c["y_sha"], c["B"])
    n_str = sum(len(sl) for sl in ys)
    n_ok = sum(sha(s) == h for sl, hl in zip(ys, c["y_sha"]) for s, h in zip(sl, hl))
    bpp = bpp_of(out["strings"], c["B"], c["H"], c["W"])
    psnr = psnr_of(x, x_hat)
    flip_free = all(f is None for f in first)
    print(f"{c['case']} q={c['quality']}: {n_ok}/{n_str} y strings identical to the reference; first diverging slice per image {first}; "
          f"bpp {bpp:.6f} (ref {c['bpp']:.6f}); psnr {psnr:.6f} dB (ref {c['psnr']:.6f}, |d| {abs(psnr - c['psnr']):.2e}; "
          f"north-star tolerance {NORTH_STAR_PSNR_TOL_DB:g} dB {'asserted' if flip_free else 'NOT met by construction: symbol flip'})")
    assert [[int(m[b].sum().item()) for b in range(c["B"])] for m in out["masks"]] == c["mask_sums"]
    if flip_free:
        assert [[len(s) for s in sl] for sl in ys] == c["y_len"]
        assert bpp == c["bpp"]
        assert abs(psnr - c["psnr"]) <= NORTH_STAR_PSNR_TOL_DB
        sub = x_hat.flatten()[::61 * 7].numpy()
        assert np.abs(sub - np.asarray(c["x_hat_sub"], np.float32)).max() <= 2e-2      # pixel level: float rounding amplified by the untrained g_s
    else:
        assert abs(bpp - c["bpp"]) <= BPP_TOL * max(1.0, c["bpp"])
        assert abs(psnr - c["psnr"]) <= PSNR_TOL_DB
        assert n_ok >= 0.5 * n_str


def _ref_string_cases():
    import json
    import os
    d = os.path.join(os.path.dirname(__file__), "golden")
    return json.load(open(os.path.join(d, "ref_strings.json"))), np.load(os.path.join(d, "ref_xhat.npz"))


@pytest.mark.parametrize("idx", range(5))
def test_decode_strings_made_by_the_reference(idx):
    """pc_codec_decompress fed the byte strings the REAL reference's compress() produced (tests/golden/ref_strings.json, made by
    tests/golden/make_golden_strings.py), image by image.  For every image whose strings this build's own encoder reproduces
    bit for bit, the reference's bytes ARE our bytes and the decoded picture must equal the reference's reconstruction to float
    rounding of the synthesis transform: PSNR within the north-star 1e-4 dB (pixels within 2e-2: with the synthetic, untrained
    weights g_s amplifies a last-bit difference of y_hat to ~5e-3).  An image with a flipped symbol cannot be decoded across
    implementations (the decoder re-derives mu / scale from what it decoded, so rANS desynchronises from the flip on): the decoder
    must then return either a picture or a clean error status -- never crash -- and the case is reported, not hidden
    (s_b3_64x128_q5 image 0 is such an image)."""
    from progressivecodec_amd._lib import PcodecError
    meta, xh = _ref_string_cases()
    c = meta[idx]
    x = inputs(c["B"], c["H"], c["W"], c["seed"], c["kind"])
    pad, unpad = compute_padding(c["H"], c["W"])
    net = gpu_codec()
    own = net.compress(F.pad(x, pad).cuda(), c["quality"], "point-based-std")
    ys = [[bytes.fromhex(h) for h in sl] for sl in c["y_hex"]]
    zs = [bytes.fromhex(h) for h in c["z_hex"]]
    assert own["strings"][1] == zs, "hyper-latent strings must match the reference exactly"
    same = [all(own["strings"][0][s][b] == ys[s][b] for s in range(len(ys))) for b in range(c["B"])]
    ref = torch.from_numpy(xh[c["case"]])
    for b in range(c["B"]):
        try:
            dec = net.decompress([[[sl[b]] for sl in ys], [zs[b]]], torch.Size(c["shape"]), c["quality"], "point-based-std")["x_hat"].cpu()
        except PcodecError as e:                   # a desynchronised stream may run out of bytes: clean status, no crash
            assert not same[b], f"decoding the reference's strings of image {b} failed although every string is identical: {e}"
            print(f"{c['case']} image {b}: a symbol flipped against the reference; decoding the foreign stream ended with: {e}")
            continue
        x_hat = F.pad(dec, unpad).clamp_(0, 1)[0]
        p_gpu = -10.0 * math.log10(torch.mean((x[b] - x_hat) ** 2).item())
        d = abs(p_gpu - c["psnr_per_image"][b])
        print(f"{c['case']} image {b}: strings identical to the reference: {same[b]}; PSNR of the GPU decode of the reference's bytes "
              f"{p_gpu:.6f} dB vs reference {c['psnr_per_image'][b]:.6f} (|d| {d:.2e}), max pixel difference {(x_hat - ref[b]).abs().max().item():.2e}")
        if same[b]:
            assert d <= NORTH_STAR_PSNR_TOL_DB
            assert (x_hat - ref[b]).abs().max().item() <= 2e-2
    assert any(same), "no image of this case is flip-free: pick another golden"


def test_config2_shape_b8_bit_exact_vs_oracle():
    """BASELINE Config 2's shape at B = 8 (the first eight of the bench's 32 crops: same generator, 256x256, q = 0.5): every byte
    string, every symbol / index / mask and x_hat equal the contract oracle's, bit for bit."""
    B, q = 8, 0.5
    x = inputs(B, 256, 256, 1)
    net = gpu_codec()
    out = net.compress(x.cuda(), q, "point-based-std")
    sym = net.read_tap("sym", np.int32).reshape(20, B, 32, 256)
    idx = net.read_tap("idx", np.int32).reshape(20, B, 32, 256)
    orc = oracle_codec("cdet")
    taps = {}
    ref = orc.compress(x, q, taps=taps)
    assert out["strings"][1] == ref["strings"][1]
    for s, (a, b) in enumerate(zip(out["strings"][0], ref["strings"][0])):
        assert a == b, f"y strings of slice {s} differ"
    for i in range(10):
        for key, step in ((f"b{i}", i), (f"e{i}", 10 + i)):
            assert np.array_equal(sym[step], taps[key]["sym"].numpy().reshape(B, 32, 256)), key
            assert np.array_equal(idx[step], taps[key]["idx"].numpy().reshape(B, 32, 256)), key
    for m, rm in zip(out["masks"], ref["masks"]):
        assert np.array_equal(m.cpu().numpy(), rm.numpy())
    dec = net.decompress(out["strings"], out["shape"], q, "point-based-std")["x_hat"].cpu()
    rdec = orc.decompress(ref["strings"], ref["shape"], q)["x_hat"]
    assert np.array_equal(dec.numpy().view(np.uint32), rdec.numpy().view(np.uint32))


def test_batch_invariance_and_roundtrip_at_bench_size():
    """Config 2 shape (256x256 crops): an image codes identically alone and inside a batch; decode(encode(x))
    equals the encoder's own reconstruction of the latents (x_hat identical for B=1 and batch)."""
    net = gpu_codec()
    x = inputs(4, 256, 256, 21).cuda()
    q = 0.5
    out = net.compress(x, q, "point-based-std")
    one = net.compress(x[2:3].contiguous(), q, "point-based-std")
    for sl_b, sl_1 in zip(out["strings"][0], one["strings"][0]):
        assert sl_b[2] == sl_1[0]
    assert out["strings"][1][2] == one["strings"][1][0]
    dec = net.decompress(out["strings"], out["shape"], q, "point-based-std")["x_hat"]
    dec1 = net.decompress(one["strings"], one["shape"], q, "point-based-std")["x_hat"]
    assert torch.equal(dec[2], dec1[0])
    assert dec.min().item() >= 0.0 and dec.max().item() <= 1.0
    h = w = 16
    k = int(32 * h * w)
    for m in out["masks"]:
        s = m.sum(dim=(1, 2, 3)).cpu()
        assert ((s - 0.05 * k).abs() <= 2).all(), s      # top 5 % per image (ties may add a few)


def test_quality_levels_monotone_bytes():
    """More of the enhancement latent is coded as the mask level grows (train.py:293 level list, subset)."""
    net = gpu_codec()
    x = inputs(1, 128, 128, 5).cuda()
    sizes = []
    for q in (0, 0.05, 0.5, 2, 5, 10):
        out = net.compress(x, q, "point-based-std")
        sizes.append(sum(len(s) for sl in out["strings"][0] for s in sl))
        dec = net.decompress(out["strings"], out["shape"], q, "point-based-std")["x_hat"]
        assert dec.shape == x.shape
    assert sizes == sorted(sizes), sizes


def test_error_behaviour():
    net = gpu_codec()
    with pytest.raises(ValueError):
        net.compress(torch.rand(1, 3, 100, 128).cuda(), 0.5)
    with pytest.raises(NotImplementedError):
        net.compress(torch.rand(1, 3, 64, 64).cuda(), 0.5, mask_pol="random")
    out = net.compress(torch.rand(1, 3, 64, 64).cuda(), 0.0, "point-based-std")
    bad = [[s[0][:8]] for s in out["strings"][0]]
    from progressivecodec_amd._lib import PcodecError
    with pytest.raises(PcodecError):
        net.decompress([bad, out["strings"][1]], out["shape"], 0.0, "point-based-std")


def test_compress_with_ac_harness_rd_table():
    """The harness of training/step.py:277-404 over the authors' level list (train.py:293) on two images (one needs
    padding): the GPU RD table equals the oracle's (same harness restated in oracle/codec_ref.py) to the last bit of
    every byte count, and PSNR to float rounding of the final mean."""
    from progressivecodec_amd.harness import PR_LIST, compress_with_ac
    levels = [PR_LIST[i] for i in (0, 1, 4, 9, 12)]
    imgs = [inputs(1, 64, 128, 31), inputs(1, 96, 72, 32, "smooth")]
    bpp, psnr, dec_t, rows = compress_with_ac(gpu_codec(), imgs, levels)
    orc = oracle_codec("cdet")
    k = 0
    for x in imgs:
        pad, unpad = compute_padding(x.shape[2], x.shape[3])
        xp = F.pad(x, pad)
        for q in levels:
            o = orc.compress(xp, q)
            xh = F.pad(orc.decompress(o["strings"], o["shape"], q)["x_hat"], unpad).clamp_(0, 1)
            assert rows[k]["bpp"] == bpp_of(o["strings"], 1, x.shape[2], x.shape[3])
            assert abs(rows[k]["psnr"] - psnr_of(x, xh)) < 1e-5
            k += 1
    assert bpp == sorted(bpp)


def test_config1_shape_bit_exact_vs_oracle():
    """BASELINE.json Config 1 shape (one 512x768 image, one mask level): every byte string and x_hat equal the oracle's."""
    x = inputs(1, 512, 768, 100, "smooth")
    net = gpu_codec()
    q = 0.5
    out = net.compress(x.cuda(), q, "point-based-std")
    orc = oracle_codec("cdet")
    ref = orc.compress(x, q)
    assert out["strings"][1] == ref["strings"][1]
    assert out["strings"][0] == ref["strings"][0]
    assert tuple(out["shape"]) == (8, 12)
    dec = net.decompress(out["strings"], out["shape"], q, "point-based-std")["x_hat"].cpu()
    rdec = orc.decompress(ref["strings"], ref["shape"], q)["x_hat"]
    assert np.array_equal(dec.numpy().view(np.uint32), rdec.numpy().view(np.uint32))


def test_large_tiles_round_trip_properties():
    """Config 4 tile size (1024x1024), 2 images: size-independent properties -- the decoder reproduces exactly the
    reconstruction implied by the encoder's own symbols for every image independently (image 1 alone == image 1 in the
    batch), byte counts are plausible, x_hat is inside [0,1]."""
    net = gpu_codec()
    x = inputs(2, 1024, 1024, 77).cuda()
    q = 2
    out = net.compress(x, q, "point-based-std")
    one = net.compress(x[1:2].contiguous(), q, "point-based-std")
    assert [sl[1] for sl in out["strings"][0]] == [sl[0] for sl in one["strings"][0]]
    dec = net.decompress(out["strings"], out["shape"], q, "point-based-std")["x_hat"]
    dec1 = net.decompress(one["strings"], one["shape"], q, "point-based-std")["x_hat"]
    assert torch.equal(dec[1], dec1[0])
    assert 0.0 <= dec.min().item() and dec.max().item() <= 1.0
    k = 32 * 64 * 64
    for m in out["masks"]:
        s = m.sum(dim=(1, 2, 3)).cpu()
        assert ((s - 0.2 * k).abs() <= 4).all(), s
    bpp = bpp_of(out["strings"], 2, 1024, 1024)
    assert 1.0 < bpp < 10.0


def test_shared_base_levels_equal_per_level_calls():
    """compress_levels / decompress_levels compute the level-independent part once; every
    string, mask and x_hat must equal, bit for bit, what one compress()/decompress() call per level returns."""
    net = gpu_codec()
    x = inputs(2, 128, 192, 55).cuda()
    levels = [0, 0.05, 0.5, 0, 2, 10]
    datas = net.compress_levels(x, levels, "point-based-std")
    assert len(datas) == len(levels)
    singles = [net.compress(x, q, "point-based-std") for q in levels]
    for q, d, s in zip(levels, datas, singles):
        assert d["strings"][1] == s["strings"][1], f"z strings differ at level {q}"
        assert d["strings"][0] == s["strings"][0], f"y strings differ at level {q}"
        assert tuple(d["shape"]) == tuple(s["shape"])
        assert len(d["masks"]) == len(s["masks"])
        for m, ms in zip(d["masks"], s["masks"]):
            assert torch.equal(m, ms)
    assert all(d["strings"][0][i] is datas[0]["strings"][0][i] for d in datas for i in range(10))   # base stored once
    outs = net.decompress_levels([d["strings"] for d in datas], datas[0]["shape"], levels, "point-based-std")
    for q, o, s in zip(levels, outs, singles):
        ref = net.decompress(s["strings"], s["shape"], q, "point-based-std")["x_hat"]
        assert torch.equal(o["x_hat"], ref), f"x_hat differs at level {q}"
    # a subset / different order of levels decodes to the same pictures
    sub = [4, 1]
    outs2 = net.decompress_levels([datas[i]["strings"] for i in sub], datas[0]["shape"], [levels[i] for i in sub], "point-based-std")
    for j, i in enumerate(sub):
        assert torch.equal(outs2[j]["x_hat"], outs[i]["x_hat"])


def test_container_round_trip_through_the_codec():
    from progressivecodec_amd import container as ct
    net = gpu_codec()
    x = inputs(2, 64, 128, 56).cuda()
    levels = [0, 0.5, 3]
    datas = net.compress_levels(x, levels, "point-based-std")
    outs = net.decompress_levels([d["strings"] for d in datas], datas[0]["shape"], levels, "point-based-std")
    for b in range(2):
        buf = ct.pack([d["strings"] for d in datas], datas[0]["shape"], levels, (64, 128), image_index=b)
        strings, shape, q, size, mp = ct.unpack(buf)
        dec = net.decompress_levels(strings, shape, q, mp)
        for lv in range(len(levels)):
            assert torch.equal(dec[lv]["x_hat"][0], outs[lv]["x_hat"][b])
            assert ct.payload_bytes(buf, lv) == sum(len(sl[b]) for sl in datas[lv]["strings"][0]) + len(datas[lv]["strings"][1][b])
        one, shape1, q1, _, _ = ct.unpack(buf, [2])           # a reader that wants the last level only
        assert torch.equal(net.decompress(one[0], shape1, q1[0], mp)["x_hat"][0], outs[2]["x_hat"][b])


def test_harness_shared_base_gives_the_same_rd_table():
    from progressivecodec_amd.harness import PR_LIST, compress_with_ac
    imgs = [inputs(1, 64, 128, 31), inputs(1, 96, 72, 32, "smooth")]
    a = compress_with_ac(gpu_codec(), imgs, PR_LIST)
    b = compress_with_ac(gpu_codec(), imgs, PR_LIST, shared_base=True)
    assert a[0] == b[0] and a[1] == b[1]
    assert [(r["quality"], r["bpp"], r["psnr"]) for r in a[3]] == [(r["quality"], r["bpp"], r["psnr"]) for r in b[3]]


# ------------------------------------------------------------------ likelihood path
LIK_RTOL = 3e-7     # one float32 ulp: the HIP kernel and the oracle round a double erfc from different libms (ocml / scipy)


@pytest.mark.parametrize("case", [(2, 64, 64, 11, "rand", 0), (2, 64, 64, 11, "rand", 0.5), (1, 128, 128, 12, "smooth", 2), (1, 64, 192, 13, "rand", 10)])
def test_forward_single_quality_vs_oracle(case):
    """pc_codec_forward against the oracle's restatement in the contract back-end: x_hat bit-exact (and equal to
    decompress(compress(x))), likelihoods within one float32 ulp, estimated bits to 1e-7 relative."""
    B, H, W, seed, kind, q = case
    x = inputs(B, H, W, seed, kind)
    net = gpu_codec()
    out = net.forward_single_quality(x.cuda(), q, "point-based-std")
    ref = oracle_codec("cdet").forward_single_quality(x, q)
    ly, lz = out["likelihoods"]["y"].cpu(), out["likelihoods"]["z"].cpu()
    ry, rz = ref["likelihoods"]["y"], ref["likelihoods"]["z"]
    assert ly.shape == ry.shape and lz.shape == rz.shape
    assert np.array_equal(out["x_hat"].cpu().numpy().view(np.uint32), ref["x_hat"].numpy().view(np.uint32))
    rel = ((ly - ry).abs() / ry).max().item()
    assert rel <= LIK_RTOL, rel
    assert ((lz - rz).abs() / rz).max().item() <= 5e-6
    bits = lambda t: float(-torch.log2(t.double()).sum())
    assert abs(bits(ly) - bits(ry)) <= 1e-7 * bits(ry)
    assert abs(bits(lz) - bits(rz)) <= 1e-6 * bits(rz)
    d = net.compress(x.cuda(), q, "point-based-std")
    xh = net.decompress(d["strings"], d["shape"], q, "point-based-std")["x_hat"]
    assert torch.equal(out["x_hat"], xh)
    for m, ms in zip(out["masks"], d["masks"]):
        assert torch.equal(m, ms)


def test_forward_single_quality_vs_reference_goldens():
    """Against forward_single_quality of the real reference (tests/golden/forward.npz): estimated bits within 2e-3 relative
    (a symbol that flips under float rounding moves its own likelihood), hyper-latent bits within 1e-5, PSNR within 2e-3 dB."""
    import json
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "forward.npz"))
    net = gpu_codec()
    for m in json.loads(bytes(g["meta_json"]).decode()):
        x = inputs(m["B"], m["H"], m["W"], m["seed"], m["kind"])
        out = net.forward_single_quality(x.cuda(), m["quality"], "point-based-std")
        ly, lz = out["likelihoods"]["y"].cpu(), out["likelihoods"]["z"].cpu()
        assert list(ly.shape) == m["y_shape"] and list(lz.shape) == m["z_shape"]
        by, bz = float(-torch.log2(ly.double()).sum()), float(-torch.log2(lz.double()).sum())
        key = f"{m['case']}_q{m['quality']}"
        sub = ly.flatten()[::53].numpy()
        frac_close = float(np.mean(np.abs(sub - g[key + "|y_sub"]) <= 1e-5 * g[key + "|y_sub"]))
        print(f"{key}: bits y {by:.3f} (ref {m['bits_y']:.3f}), z {bz:.3f} (ref {m['bits_z']:.3f}); {100 * frac_close:.1f} % of sampled likelihoods within 1e-5")
        assert abs(by - m["bits_y"]) <= 2e-3 * m["bits_y"]
        assert abs(bz - m["bits_z"]) <= 1e-5 * m["bits_z"]
        assert frac_close >= 0.98
        assert abs(psnr_of(x, out["x_hat"].cpu()) - m["psnr"]) <= PSNR_TOL_DB


def test_estimated_rate_tracks_the_coded_rate():
    """estimate_rd (test_epoch's table, step.py:215-267) against compress_with_ac on the same images: the likelihood-based bpp is
    the model entropy at the continuous scale, the coder works from the 64-entry scale table (and the synthetic, untrained
    weights fit the data loosely) -- the two agree within 6 %."""
    from progressivecodec_amd.harness import compress_with_ac, estimate_rd
    imgs = [inputs(1, 128, 128, 41), inputs(1, 64, 192, 42, "smooth")]
    levels = [0, 0.5, 3, 10]
    est_bpp, est_psnr, _ = estimate_rd(gpu_codec(), imgs, levels)
    bpp, psnr, _, _ = compress_with_ac(gpu_codec(), imgs, levels, shared_base=True)
    for e, c in zip(est_bpp, bpp):
        assert 0.94 * e <= c <= 1.06 * e, (e, c)
    assert np.allclose(est_psnr, psnr, atol=1e-9)
    assert est_bpp == sorted(est_bpp)


@pytest.mark.parametrize("pol,q", [("two-levels", 0), ("two-levels", 1), ("three-levels-std", 0), ("three-levels-std", 1), ("three-levels-std", 2)])
def test_other_mask_policies_bit_exact_vs_oracle(pol, q):
    """SURVEY section 8(f) rank 4 (the policies that need no learned or gradient map): layers/masking.py:227-247."""
    x = inputs(2, 64, 128, 77)
    net = gpu_codec()
    out = net.compress(x.cuda(), q, pol)
    orc = oracle_codec("cdet")
    ref = orc.compress(x, q, pol)
    assert out["strings"][1] == ref["strings"][1] and out["strings"][0] == ref["strings"][0]
    for m, rm in zip(out["masks"], ref["masks"]):
        assert np.array_equal(m.cpu().numpy(), rm.numpy())
    if pol == "three-levels-std" and q == 1:
        k = 32 * 4 * 8
        assert all(abs(int(m[b].sum().item()) - 0.2 * k) <= 2 for m in out["masks"] for b in range(2))    # top 20 % by scale
    dec = net.decompress(out["strings"], out["shape"], q, pol)["x_hat"].cpu()
    rdec = orc.decompress(ref["strings"], ref["shape"], q, pol)["x_hat"]
    assert np.array_equal(dec.numpy().view(np.uint32), rdec.numpy().view(np.uint32))


def test_4k_frame_two_levels_properties():
    """BASELINE Config 5 shape: one 3840x2160 frame (centre-padded to 3840x2176 as training/step.py:318 does), two mask levels
    through the shared-base path.  Too large for the oracle in seconds, so size-independent properties: the multi-level result
    equals the per-level calls, decode reproduces the encoder's reconstruction path (forward_single_quality's x_hat), masks hold
    the requested share, byte counts grow with the level."""
    from progressivecodec_amd.harness import compute_padding as cp
    net = gpu_codec()
    g = torch.Generator().manual_seed(5)
    lo_res = torch.rand(1, 3, 270, 480, generator=g)
    x = F.interpolate(lo_res, size=(2160, 3840), mode="bilinear", align_corners=False).clamp(0, 1)
    pad, unpad = cp(2160, 3840)
    xp = F.pad(x, pad).cuda()
    assert tuple(xp.shape[2:]) == (2176, 3840)
    levels = [0.5, 3]
    datas = net.compress_levels(xp, levels, "point-based-std")
    one = net.compress(xp, 3, "point-based-std")
    assert datas[1]["strings"] == one["strings"]
    outs = net.decompress_levels([d["strings"] for d in datas], datas[0]["shape"], levels, "point-based-std")
    fwd = net.forward_single_quality(xp, 3, "point-based-std")
    assert torch.equal(outs[1]["x_hat"], fwd["x_hat"])
    k = 32 * 136 * 240
    for lv, share in zip(range(2), (0.05, 0.3)):
        for m in datas[lv]["masks"]:
            assert abs(int(m.sum().item()) - share * k) <= 8
    nbytes = [sum(len(s[0]) for s in d["strings"][0]) for d in datas]
    assert nbytes[0] < nbytes[1]
    x_hat = F.pad(outs[1]["x_hat"], unpad)
    assert tuple(x_hat.shape) == (1, 3, 2160, 3840) and 0.0 <= x_hat.min().item() and x_hat.max().item() <= 1.0
    bits = float(-torch.log2(fwd["likelihoods"]["y"].double()).sum() - torch.log2(fwd["likelihoods"]["z"].double()).sum())
    coded = 8 * (nbytes[1] + len(datas[1]["strings"][1][0]))
    assert 0.9 * bits <= coded <= 1.1 * bits


def test_encoder_and_decoder_objects_side_by_side():
    """bench.py's default schedule: a second codec object decodes batch i on its own stream and host thread while the first one
    encodes batch i+1.  Every string and every reconstruction must equal what the same calls give one after the other, and the decoder
    object must decode the encoder object's streams (same weights, same contract)."""
    import queue
    import threading
    from progressivecodec_amd import ChannelProgresssiveWACNN
    from tests.util import synth_sd
    enc = gpu_codec()
    dec = ChannelProgresssiveWACNN(device="cuda:0")          # a second object of its own (gpu_codec() is cached: one object is not re-entrant)
    dec.load_state_dict(synth_sd())
    assert dec._h.value != enc._h.value
    g = torch.Generator().manual_seed(77)
    batches = [torch.rand(4, 3, 128, 128, generator=g).cuda() for _ in range(6)]
    q = 0.5
    want = []
    for x in batches:                                           # the sequential answer, on the encoder object alone
        o = enc.compress(x, q, "point-based-std")
        want.append((o["strings"], enc.decompress(o["strings"], o["shape"], q, "point-based-std")["x_hat"].clone()))
    qu, got, err = queue.Queue(maxsize=2), [], []
    s_enc, s_dec = torch.cuda.Stream(), torch.cuda.Stream()

    def decoder():
        try:
            torch.cuda.set_device(0)
            with torch.cuda.stream(s_dec):
                while True:
                    o = qu.get()
                    if o is None:
                        return
                    got.append(dec.decompress(o["strings"], o["shape"], q, "point-based-std")["x_hat"].clone())
        except BaseException as e:                              # noqa: BLE001
            err.append(e)
            while qu.get() is not None:
                pass

    th = threading.Thread(target=decoder)
    th.start()
    outs = []
    with torch.cuda.stream(s_enc):
        for rep in range(2):
            for x in batches:
                o = enc.compress(x, q, "point-based-std")
                outs.append(o)
                qu.put(o)
    qu.put(None)
    th.join(timeout=300)
    assert not th.is_alive() and not err, err
    s_enc.synchronize(); s_dec.synchronize()
    assert len(got) == 12
    for i, (o, xh) in enumerate(zip(outs, got)):
        strings, x_want = want[i % 6]
        assert o["strings"] == strings
        assert torch.equal(xh, x_want)


def test_one_object_from_two_threads():
    """One object is not re-entrant (its workspaces, streams and result strings belong to the call in progress).  The Python mirror
    serialises calls per object, so two threads get the right answers; at the C ABI a caller that finds the object busy gets
    PC_ERR_STATE.  Nothing is silently corrupted either way."""
    import ctypes as C
    import threading
    from progressivecodec_amd._lib import lib
    net = gpu_codec()
    g = torch.Generator().manual_seed(91)
    xs = [torch.rand(2, 3, 128, 128, generator=g).cuda() for _ in range(2)]
    want = [net.compress(x, 0.5, "point-based-std")["strings"] for x in xs]
    good, bad = [0], []

    def worker(i):
        torch.cuda.set_device(0)
        for _ in range(15):
            s = net.compress(xs[i], 0.5, "point-based-std")["strings"]
            if s == want[i]:
                good[0] += 1
            else:
                bad.append("wrong strings")

    ths = [threading.Thread(target=worker, args=(i,)) for i in range(2)]
    for t in ths:
        t.start()
    for t in ths:
        t.join(timeout=300)
    assert not bad and good[0] == 30, (good, bad[:3])
    # the C ABI underneath, without the mirror's lock
    rcs = []

    def raw(i):
        torch.cuda.set_device(0)
        for _ in range(15):
            rcs.append(lib().pc_codec_compress(net._h, C.c_void_p(xs[i].data_ptr()), 2, 128, 128, 0.5, 0, None, None))

    ths = [threading.Thread(target=raw, args=(i,)) for i in range(2)]
    for t in ths:
        t.start()
    for t in ths:
        t.join(timeout=300)
    torch.cuda.synchronize()
    assert set(rcs) <= {0, -8} and rcs.count(0) > 0, rcs
    print(f"same object, two threads at the C ABI: {rcs.count(0)} calls served, {rcs.count(-8)} refused with PC_ERR_STATE")
    assert net.compress(xs[0], 0.5, "point-based-std")["strings"] == want[0]


def test_config5_4k_frame_eight_levels_and_gather():
    """BASELINE Config 5 on one rank: one 3840x2160 frame (padded to 3840x2176), EIGHT progressive levels through the shared-base
    path, every level decoded, then the bitstream gather of bench.py over the process group (world size 1 here: RCCL on the GPU
    box; the 8-rank form of the same code runs over gloo in tests/test_parallel.py).  Properties: each level's strings equal the
    per-level compress() call's; byte counts grow with the level; the mask shares are the requested quantiles (the multi-block
    quantile path: 32 x 136 x 240 = 1.04 M keys per slice); every level decodes into [0, 1]; the gathered strings are the local ones."""
    import torch.distributed as dist
    from progressivecodec_amd.harness import compute_padding as cp
    from progressivecodec_amd.parallel import gather_bitstreams
    net = gpu_codec()
    g = torch.Generator().manual_seed(55)
    lo_res = torch.rand(1, 3, 270, 480, generator=g)
    x = (F.interpolate(lo_res, size=(2160, 3840), mode="bilinear", align_corners=False) + 0.03 * torch.randn(1, 3, 2160, 3840, generator=g)).clamp(0, 1)
    pad, unpad = cp(2160, 3840)
    xp = F.pad(x, pad).cuda()
    levels = [0.05, 0.25, 0.5, 1, 2, 3, 5, 10]
    datas = net.compress_levels(xp, levels, "point-based-std")
    assert len(datas) == 8
    for lv in (0, 4, 7):
        one = net.compress(xp, levels[lv], "point-based-std")
        assert datas[lv]["strings"] == one["strings"]
    nbytes = [sum(len(s[0]) for s in d["strings"][0]) for d in datas]
    assert all(a < b for a, b in zip(nbytes, nbytes[1:])), nbytes
    k = 32 * 136 * 240
    for lv, q in enumerate(levels):
        for m in datas[lv]["masks"]:
            assert abs(int(m.sum().item()) - min(1.0, q * 0.1) * k) <= 8
    outs = net.decompress_levels([d["strings"] for d in datas], datas[0]["shape"], levels, "point-based-std")
    psnr = []
    for o in outs:
        xh = F.pad(o["x_hat"], unpad)
        assert tuple(xh.shape) == (1, 3, 2160, 3840) and 0.0 <= xh.min().item() and xh.max().item() <= 1.0
        psnr.append(float(-10 * torch.log10(((xh.cpu() - x) ** 2).mean())))
    fwd = net.forward_single_quality(xp, levels[-1], "point-based-std")
    assert torch.equal(outs[-1]["x_hat"], fwd["x_hat"])
    print(f"Config 5 frame: y bytes per level {nbytes}, PSNR per level (synthetic weights) {[round(p, 2) for p in psnr]}")
    # levels 0.5 and 10 against the REAL reference's fixture of the same frame (tests/golden/config45.json)
    for c in _golden_json("config45.json")["cases"]:
        if c["config"] != "Config 5":
            continue
        lv = levels.index(c["quality"])
        _check_against_reference_case(c, datas[lv]["strings"], x, F.pad(outs[lv]["x_hat"], unpad).cpu().clamp(0, 1), datas[lv]["masks"], "Config 5 frame")
    own = not dist.is_initialized()
    if own:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29561")
        dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        for d in (datas[0], datas[7]):
            assert gather_bitstreams(d["strings"][0]) == d["strings"][0]
            assert gather_bitstreams([d["strings"][1]]) == [d["strings"][1]]
    finally:
        if own:
            dist.destroy_process_group()


@pytest.mark.parametrize("idx", [0, 1, 2])
def test_cust_map_bit_exact_vs_oracle_and_reference_goldens(idx):
    """compress()/decompress() with cust_map (CHProg_cnn.py:721-722,823 -> masking.py:171-194): GPU == oracle (contract back-end) on
    every string, mask and x_hat; mask popcounts equal the reference's, bpp / PSNR within the tolerances of the other goldens."""
    import json
    import os
    c = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "cust_map.json")))[idx]
    x = inputs(c["B"], c["H"], c["W"], c["seed"], c["kind"])
    cm = torch.rand(c["B"], 320, c["H"] // 16, c["W"] // 16, generator=torch.Generator().manual_seed(c["seed"] + 1000))
    net = gpu_codec()
    out = net.compress(x.cuda(), c["quality"], c["mask_pol"], cust_map=cm)
    orc = oracle_codec("cdet")
    ref = orc.compress(x, c["quality"], c["mask_pol"], cust_map=cm)
    assert out["strings"][0] == ref["strings"][0] and out["strings"][1] == ref["strings"][1]
    for m, rm in zip(out["masks"], ref["masks"]):
        assert np.array_equal(m.cpu().numpy(), rm.numpy())
    assert [[int(m[b].sum().item()) for b in range(c["B"])] for m in out["masks"]] == c["mask_sums"]
    dec = net.decompress(out["strings"], out["shape"], c["quality"], c["mask_pol"], cust_map=cm)["x_hat"].cpu()
    rdec = orc.decompress(ref["strings"], ref["shape"], c["quality"], c["mask_pol"], cust_map=cm)["x_hat"]
    assert np.array_equal(dec.numpy().view(np.uint32), rdec.numpy().view(np.uint32))
    assert abs(bpp_of(out["strings"], c["B"], c["H"], c["W"]) - c["bpp"]) <= BPP_TOL * max(1.0, c["bpp"])
    assert abs(psnr_of(x, dec.clamp(0, 1)) - c["psnr"]) <= PSNR_TOL_DB
    # without the map the masks differ (the map really is what was thresholded)
    if 0 < c["quality"] < 10 and c["mask_pol"] == "point-based-std":
        plain = net.compress(x.cuda(), c["quality"], c["mask_pol"])
        assert any(not torch.equal(a, b) for a, b in zip(plain["masks"], out["masks"]))


def test_error_behaviour_of_the_wider_entry_points():
    from progressivecodec_amd._lib import PcodecError
    net = gpu_codec()
    x = torch.rand(1, 3, 64, 64).cuda()
    with pytest.raises(ValueError):
        net.compress_levels(x, [], "point-based-std")
    with pytest.raises(ValueError):
        net.compress_levels(torch.rand(1, 3, 100, 64).cuda(), [0.5])
    with pytest.raises(NotImplementedError):
        net.forward_single_quality(x, 0.5, training=True)
    with pytest.raises(NotImplementedError):
        net.compress(x, 0.5, mask_pol="three-levels-learnable")
    with pytest.raises(ValueError):
        net.compress(x, 0.5, cust_map=torch.rand(1, 320, 8, 8))          # wrong map shape
    d = net.compress_levels(x, [0, 0.5])
    with pytest.raises(ValueError):
        net.decompress_levels([d[0]["strings"]], d[0]["shape"], [0, 0.5])   # one entry per level
    with pytest.raises(ValueError):
        net.decompress_levels([d[0]["strings"], d[0]["strings"]], d[0]["shape"], [0, 0.5])   # level 0.5 without enhancement strings
    bad = [[s[0][:6]] for s in d[1]["strings"][0]]
    with pytest.raises(PcodecError):
        net.decompress_levels([d[0]["strings"], [bad, d[1]["strings"][1]]], d[0]["shape"], [0, 0.5])


# ------------------------------------------------------------------ multiple_encoder=True and force_enhanced (VERDICT r01 missing 2 / 6)
def _multienc_gpu():
    import functools
    from tests.test_oracle_vs_golden import multienc_sd
    global _ME_NET
    try:
        return _ME_NET
    except NameError:
        from progressivecodec_amd import ChannelProgresssiveWACNN
        _ME_NET = ChannelProgresssiveWACNN(device="cuda:0", multiple_encoder=True)
        _ME_NET.load_state_dict(multienc_sd())
        return _ME_NET


@pytest.mark.parametrize("idx", [0, 1, 2])
def test_multiple_encoder_bit_exact_vs_oracle_and_reference_goldens(idx):
    """Two analysis transforms, y = cat(g_a[0](x), g_a[1](x)) (CHProg_cnn.py:131-144,691-697): GPU == contract oracle on every
    string, mask and x_hat; hyper-latent strings and mask popcounts equal the REAL reference's (tests/golden/multienc.json), bpp / PSNR
    within the flip tolerances (1e-4 dB when no y string differs)."""
    import json
    import os
    from oracle.codec_ref import RefCodec
    from tests.test_oracle_vs_golden import multienc_sd
    c = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "multienc.json")))["multienc"][idx]
    x = inputs(c["B"], c["H"], c["W"], c["seed"], c["kind"])
    net = _multienc_gpu()
    out = net.compress(x.cuda(), c["quality"], "point-based-std")
    orc = RefCodec(multienc_sd(), "cdet")
    ref = orc.compress(x, c["quality"])
    assert out["strings"][1] == ref["strings"][1] and out["strings"][0] == ref["strings"][0]
    for m, rm in zip(out["masks"], ref["masks"]):
        assert np.array_equal(m.cpu().numpy(), rm.numpy())
    dec = net.decompress(out["strings"], out["shape"], c["quality"], "point-based-std")["x_hat"].cpu()
    rdec = orc.decompress(ref["strings"], ref["shape"], c["quality"])["x_hat"]
    assert np.array_equal(dec.numpy().view(np.uint32), rdec.numpy().view(np.uint32))
    assert [sha(s) for s in out["strings"][1]] == c["z_sha"]
    assert [[int(m[b].sum().item()) for b in range(c["B"])] for m in out["masks"]] == c["mask_sums"]
    first = flip_report(out["strings"][0], c["y_sha"], c["B"])
    psnr = psnr_of(x, dec.clamp(0, 1))
    print(f"{c['case']} q={c['quality']}: first diverging slice per image {first}; psnr {psnr:.6f} (ref {c['psnr']:.6f})")
    assert abs(psnr - c["psnr"]) <= (NORTH_STAR_PSNR_TOL_DB if all(f is None for f in first) else PSNR_TOL_DB)
    assert abs(bpp_of(out["strings"], c["B"], c["H"], c["W"]) - c["bpp"]) <= (0 if all(f is None for f in first) else BPP_TOL * c["bpp"])


@pytest.mark.parametrize("idx", [0, 1])
def test_force_enhanced_vs_oracle_and_reference_goldens(idx):
    """forward_single_quality(quality=0, force_enhanced=True) (CHProg_cnn.py:1006,1022,1064): x_hat bit-exact vs the contract oracle,
    likelihoods within one float32 ulp; estimated bits within 2e-3 of the REAL reference's, 640 likelihood channels, all-zero masks."""
    import json
    import os
    c = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "multienc.json")))["forced"][idx]
    x = inputs(c["B"], c["H"], c["W"], c["seed"], c["kind"])
    net = gpu_codec()
    out = net.forward_single_quality(x.cuda(), 0, "point-based-std", force_enhanced=True)
    ref = oracle_codec("cdet").forward_single_quality(x, 0, force_enhanced=True)
    ly, ry = out["likelihoods"]["y"].cpu(), ref["likelihoods"]["y"]
    assert list(ly.shape) == c["y_shape"]
    assert np.array_equal(out["x_hat"].cpu().numpy().view(np.uint32), ref["x_hat"].numpy().view(np.uint32))
    assert ((ly - ry).abs() / ry).max().item() <= LIK_RTOL
    assert all(int(m.sum().item()) == 0 for m in out["masks"]) and len(out["masks"]) == 10
    by = float(-torch.log2(ly.double()).sum())
    assert abs(by - c["bits_y"]) <= 2e-3 * c["bits_y"]
    assert abs(psnr_of(x, out["x_hat"].cpu()) - c["psnr"]) <= PSNR_TOL_DB
    plain = net.forward_single_quality(x.cuda(), 0, "point-based-std")
    assert plain["likelihoods"]["y"].shape[1] == 320 and torch.equal(plain["likelihoods"]["y"], out["likelihoods"]["y"][:, :320])


def test_module_surface_and_update_scale_table():
    """nn.Module surface and update(scale_table=..., force=...) (models/cnn.py:137-142, entropy_models.py:588-597)."""
    from progressivecodec_amd import ChannelProgresssiveWACNN
    from progressivecodec_amd.model import get_scale_table
    from tests.util import synth_sd
    net = ChannelProgresssiveWACNN(device="cuda:0")
    assert isinstance(net, torch.nn.Module) and net.eval() is net
    net.load_state_dict(synth_sd())
    sd = net.state_dict()
    assert len(sd) == 1019 and torch.equal(sd["g_a.0.weight"], synth_sd()["g_a.0.weight"])
    assert sum(p.numel() for p in net.parameters()) == 152137398                 # authors' log
    assert net.update() is False                                                 # tables came with the state dict: nothing rebuilt
    x = inputs(1, 64, 64, 3).cuda()
    a = net.compress(x, 0.5, "point-based-std")
    assert net.update(force=True) is True                                        # rebuilt from get_scale_table(): identical tables
    b = net.compress(x, 0.5, "point-based-std")
    assert a["strings"] == b["strings"]
    # a coarser custom table: different indexes / strings, still decodable; the oracle with the same table agrees bit for bit
    coarse = get_scale_table(0.11, 256, 32)
    assert net.update(scale_table=coarse, force=True) is True
    c = net.compress(x, 0.5, "point-based-std")
    assert c["strings"][0] != a["strings"][0] and c["strings"][1] == a["strings"][1]
    from oracle.codec_ref import RefCodec, gaussian_conditional_tables
    sd2 = dict(synth_sd())
    sd2["gaussian_conditional.scale_table"] = coarse
    orc = RefCodec(sd2, "cdet", gc_tables=gaussian_conditional_tables(coarse))
    r = orc.compress(x.cpu(), 0.5)
    assert c["strings"][0] == r["strings"][0]
    xh = net.decompress(c["strings"], c["shape"], 0.5, "point-based-std")["x_hat"].cpu()
    assert np.array_equal(xh.numpy().view(np.uint32), orc.decompress(r["strings"], r["shape"], 0.5)["x_hat"].numpy().view(np.uint32))
    assert torch.equal(net.state_dict()["gaussian_conditional.scale_table"], coarse)


def test_quality_zero_string_count_matches_num_slices():
    """after compress at quality 0 the bulk-copy ABI reports 10*B + B strings, like pc_codec_num_slices."""
    import ctypes as C
    from progressivecodec_amd._lib import lib
    net = gpu_codec()
    B = 2
    out = net.compress(inputs(B, 64, 64, 9).cuda(), 0.0, "point-based-std")
    tot, n = C.c_size_t(), C.c_int()
    assert lib().pc_codec_strings_size(net._h, C.byref(tot), C.byref(n)) == 0
    assert lib().pc_codec_num_slices(net._h) == 10 and n.value == 10 * B + B and len(out["strings"][0]) == 10
    buf = C.create_string_buffer(tot.value)
    lens = (C.c_size_t * n.value)()
    assert lib().pc_codec_copy_strings(net._h, buf, tot.value, lens, n.value - 1) == -3         # lens too short: PC_ERR_BUFFER
    assert lib().pc_codec_copy_strings(net._h, buf, tot.value, lens, n.value) == 0


def test_row_table_cache_is_bounded_and_freed():
    """distinct image geometries must not grow HBM without bound; destroying a codec returns its memory."""
    from progressivecodec_amd import ChannelProgresssiveWACNN
    from tests.util import synth_sd
    torch.cuda.synchronize()
    net = ChannelProgresssiveWACNN(device="cuda:0")
    net.load_state_dict(synth_sd())
    x0 = torch.rand(1, 3, 64, 64).cuda()
    net.compress(x0, 0.5, "point-based-std")
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    sizes = [(64 * a, 64 * b) for a in range(1, 8) for b in range(1, 8)]          # 49 geometries, largest first below
    big = max(sizes, key=lambda s: s[0] * s[1])
    net.compress(torch.rand(1, 3, *big).cuda(), 0.5, "point-based-std")           # workspace grows to its maximum once
    torch.cuda.synchronize()
    free1 = torch.cuda.mem_get_info()[0]
    for H, W in sizes:
        o = net.compress(torch.rand(1, 3, H, W).cuda(), 0.5, "point-based-std")
        net.decompress(o["strings"], o["shape"], 0.5, "point-based-std")
    torch.cuda.synchronize()
    free2 = torch.cuda.mem_get_info()[0]
    assert free1 - free2 < 700 << 20, f"HBM grew by {(free1 - free2) >> 20} MB over 49 geometries (row-table cap is 512 MB)"
    del net
    import gc
    gc.collect()
    torch.cuda.synchronize()
    assert torch.cuda.mem_get_info()[0] >= free0, "destroying the codec must return its weights, workspace and row tables"


def test_config4_per_gpu_shard_32x1024x1024():
    """BASELINE Config 4's per-GPU shard at FULL size: 32 tiles of 1024x1024 in one call (the "~45 GB of workspace" case of DESIGN.md
    section 3; int32 pixel indices, 64-bit offsets, the multi-block quantile and the 2 GB-class buffers are all exercised).  Too large
    for the oracle, so size-independent properties: an image codes identically inside the batch and alone, the decoder reproduces
    the single-image reconstruction, masks hold the requested share, HBM use is reported."""
    net = gpu_codec()
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    g = torch.Generator().manual_seed(1000)
    x = torch.rand(32, 3, 1024, 1024, generator=g).cuda()
    q = 0.5
    out = net.compress(x, q, "point-based-std")
    torch.cuda.synchronize()
    used = (free0 - torch.cuda.mem_get_info()[0]) / 2 ** 30
    print(f"Config 4 shard (32 x 1024^2): {used:.1f} GiB of HBM beyond the weights (input batch 0.4 GiB included)")
    assert used < 120.0
    one = net.compress(x[17:18].contiguous(), q, "point-based-std")
    assert [sl[17] for sl in out["strings"][0]] == [sl[0] for sl in one["strings"][0]] and out["strings"][1][17] == one["strings"][1][0]
    k = 32 * 64 * 64
    for m in out["masks"]:
        s = m.sum(dim=(1, 2, 3)).cpu()
        assert ((s - 0.05 * k).abs() <= 4).all(), s
    dec = net.decompress(out["strings"], out["shape"], q, "point-based-std")["x_hat"]
    dec1 = net.decompress(one["strings"], one["shape"], q, "point-based-std")["x_hat"]
    assert torch.equal(dec[17], dec1[0]) and 0.0 <= dec.min().item() and dec.max().item() <= 1.0
    bpp = bpp_of(out["strings"], 32, 1024, 1024)
    assert 1.0 < bpp < 10.0
    # tiles 0 and 17 against the REAL reference's fixture (tests/golden/config45.json, make_golden_config45.py)
    xc = x.cpu()
    for c in _golden_json("config45.json")["cases"]:
        if c["config"] != "Config 4":
            continue
        i = c["index"]
        st = [[[sl[i]] for sl in out["strings"][0]], [out["strings"][1][i]]]
        _check_against_reference_case(c, st, xc[i:i + 1], dec[i:i + 1].cpu().clamp(0, 1), [m[i] for m in out["masks"]], f"Config 4 tile {i}")


# ------------------------------------------------------------------ REM model family (VERDICT r01 missing 1)
def _rem_gpu():
    global _REM_NET
    try:
        return _REM_NET
    except NameError:
        from progressivecodec_amd import ChannelProgresssiveWACNN, PostRateProcessedNetwork
        from tests.test_oracle_vs_golden import rem_post_sd
        from tests.util import synth_sd
        _REM_NET = PostRateProcessedNetwork(ChannelProgresssiveWACNN(device="cuda:0"), check_levels=[0.01, 0.25, 1.75])
        _REM_NET.load_state_dict(synth_sd(), rem_post_sd())
        return _REM_NET


@pytest.mark.parametrize("idx", range(8))
def test_rem_bit_exact_vs_oracle_and_reference_goldens(idx):
    """PostRateProcessedNetwork.compress()/decompress() (CHProgREM.py:673,896): GPU == contract oracle on every string, mask, the
    refined scales and x_hat; hyper-latent strings and shapes equal the REAL reference's (tests/golden/rem.json), bpp / PSNR within the
    flip tolerances.  Cases: below the first check level (no refinement), each of the three refinement ranges, quality 10; and
     mask_pol "two-levels" / "three-levels-std", where the block mask follows the policy but the attention mask of
    apply_latent_enhancement stays quantile-based (CHProgREM.py:385,620,832,1060)."""
    import json
    import os
    from tests.test_oracle_vs_golden import rem_oracle
    c = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "rem.json")))[idx]
    x = inputs(c["B"], c["H"], c["W"], c["seed"], c["kind"])
    net = _rem_gpu()
    pol = c.get("mask_pol", "point-based-std")
    out = net.compress(x.cuda(), c["quality"], pol)
    orc = rem_oracle("cdet")
    taps = {}
    ref = orc.compress(x, c["quality"], pol, taps=taps)
    assert out["strings"][1] == ref["strings"][1]
    for s, (a, b) in enumerate(zip(out["strings"][0], ref["strings"][0])):
        assert a == b, f"y strings of slice {s} differ"
    for m, rm in zip(out["masks"], ref["masks"]):
        assert np.array_equal(m.cpu().numpy(), rm.numpy())
    scale = net.base_net.read_tap("scale").reshape(20, c["B"], -1, 32)[13]                   # refined scale of enhancement slice 3, NHWC
    assert np.array_equal(scale.reshape(c["B"], c["H"] // 16, c["W"] // 16, 32).transpose(0, 3, 1, 2), taps["e3"]["scale"].numpy())
    dec = net.decompress(out["strings"], out["shape"], c["quality"], pol)
    rdec = orc.decompress(ref["strings"], ref["shape"], c["quality"], pol)["x_hat"]
    assert np.array_equal(dec["x_hat"].cpu().numpy().view(np.uint32), rdec.numpy().view(np.uint32))
    assert tuple(out["y_hat"].shape) == (c["B"], 320, c["H"] // 16, c["W"] // 16) and torch.equal(out["y_hat"], dec["y_hat"])
    assert [sha(s) for s in out["strings"][1]] == c["z_sha"] and list(out["shape"]) == c["shape"]
    first = flip_report(out["strings"][0], c["y_sha"], c["B"])
    flip_free = all(f is None for f in first)
    psnr = psnr_of(x, dec["x_hat"].cpu().clamp(0, 1))
    print(f"{c['case']} q={c['quality']}: first diverging slice per image {first}; psnr {psnr:.6f} (ref {c['psnr']:.6f})")
    if flip_free:
        assert [[int(m[b].sum().item()) for b in range(c["B"])] for m in out["masks"]] == c["mask_sums"]
    assert abs(psnr - c["psnr"]) <= (NORTH_STAR_PSNR_TOL_DB if flip_free else 5e-2)
    assert abs(bpp_of(out["strings"], c["B"], c["H"], c["W"]) - c["bpp"]) <= (0 if flip_free else 2e-2 * c["bpp"])
    # the REM switch is off again: the plain codec codes as before
    plain = gpu_codec().compress(x.cuda(), c["quality"], pol)
    again = net.base_net.compress(x.cuda(), c["quality"], pol)
    assert plain["strings"] == again["strings"]
    if c["quality"] > 0.01:
        assert plain["strings"][0][10:] != out["strings"][0][10:], "the refinement must change the enhancement strings"


def test_rem_is_a_module_and_reloads_from_its_own_state_dict():
    """CHProgREM.py:205: an nn.Module whose state_dict carries base_net.* and post_latent.* (the layout of the reference's REM
    checkpoints); splitting it by prefix and loading it into a fresh object codes the same bytes."""
    from progressivecodec_amd import ChannelProgresssiveWACNN, PostRateProcessedNetwork
    from progressivecodec_amd.arch import rem_param_spec
    net = _rem_gpu()
    assert isinstance(net, torch.nn.Module) and isinstance(net.base_net, torch.nn.Module) and net.eval() is net and not net.training
    sd = net.state_dict()
    base = {k[len("base_net."):]: v for k, v in sd.items() if k.startswith("base_net.")}
    post = {k[len("post_latent."):]: v for k, v in sd.items() if k.startswith("post_latent.")}
    assert len(base) + len(post) == len(sd) and set(post) == set(rem_param_spec(3, "big")) and len(base) >= 1019
    assert sum(p.numel() for p in net.parameters()) == sum(p.numel() for p in net.base_net.parameters()) + sum(v.numel() for v in post.values())
    x = inputs(1, 64, 64, 5, "smooth").cuda()
    want = net.compress(x, 0.5, "point-based-std")["strings"]
    other = PostRateProcessedNetwork(ChannelProgresssiveWACNN(device="cuda:0"), check_levels=[0.01, 0.25, 1.75])
    other.load_state_dict(base, post, strict=True)
    assert other.compress(x, 0.5, "point-based-std")["strings"] == want


def _golden_json(name):
    import json
    import os
    return json.load(open(os.path.join(os.path.dirname(__file__), "golden", name)))


def _check_against_reference_case(c, strings, x, x_hat, masks, what):
    """one image's strings / reconstruction against a fixture the reference itself made (tests/golden/config45.json): the hyper-latent
    string must be identical; flip-free -> identical bytes, mask sums and PSNR within the north-star 1e-4 dB; otherwise listed with the
    first diverging slice and held to 2e-3 relative bpp / 5e-3 dB (the bounds of the Config-2 / Config-3 tests)."""
    from progressivecodec_amd.harness import compare_with_golden_strings
    cmp_ = compare_with_golden_strings(strings, [[h] for h in c["y_sha"]], [c["z_sha"]])
    assert cmp_["z_strings_identical"] == 1, f"{what}: hyper-latent string differs from the reference's"
    bpp = bpp_of(strings, 1, c["H"], c["W"])
    psnr = psnr_of(x, x_hat)
    ff = bool(cmp_["flip_free_images"])
    print(f"{what} q={c['quality']}: y strings identical {cmp_['y_strings_identical']}/{cmp_['y_strings']}, first diverging slice {cmp_['first_diverging_slice'][0]}, "
          f"bpp {bpp:.6f} (ref {c['bpp']:.6f}), psnr {psnr:.6f} (ref {c['psnr']:.6f})")
    if ff:
        assert bpp == c["bpp"] and abs(psnr - c["psnr"]) <= NORTH_STAR_PSNR_TOL_DB
        assert [int(m.sum().item()) for m in masks] == c["mask_sums"]
    else:
        assert abs(bpp - c["bpp"]) <= BPP_TOL * max(1.0, c["bpp"]) and abs(psnr - c["psnr"]) <= 5e-3


def test_config2_b32_vs_reference_golden():
    """BASELINE.json Config 2 at FULL size against the REAL reference: bench.py's exact batch --
    torch.rand(32,3,256,256) from seed 1, quality 0.5 -- coded on the GPU and compared with tests/golden/config2.json, which
    tests/golden/make_golden_config2.py made by importing the reference in the build container (8 threads = THE golden; the other
    thread counts in the file show what a different oneDNN team does to the same strings).  Asserted: shape, the batch bpp / PSNR, and
    on every flip-free image (all 21 strings identical to the reference's) identical byte counts and PSNR within the north-star
    1e-4 dB.  Images in which float rounding flipped a symbol are listed with their first diverging slice and held to 5e-3 dB."""
    from progressivecodec_amd.harness import compare_with_golden_strings
    g = _golden_json("config2.json")
    r = g["runs"][0]
    assert r["threads"] == 8 and g["B"] == 32 and g["quality"] == 0.5
    x = torch.rand(g["B"], 3, g["H"], g["W"], generator=torch.Generator().manual_seed(g["seed"]))
    net = gpu_codec()
    out = net.compress(x.cuda(), g["quality"], g["mask_pol"])
    dec = net.decompress(out["strings"], out["shape"], g["quality"], g["mask_pol"])
    x_hat = dec["x_hat"].cpu().clamp_(0, 1)
    assert list(out["shape"]) == r["shape"] and len(out["strings"][0]) == 20
    cmp_ = compare_with_golden_strings(out["strings"], r["y_sha"], r["z_sha"])
    ys, zs = out["strings"]
    B, S = g["B"], g["H"]
    worst_ff, worst_fl = 0.0, 0.0
    for b in range(B):
        nbytes = sum(len(ys[s][b]) for s in range(20)) + len(zs[b])
        psnr = psnr_of(x[b], x_hat[b])
        d = abs(psnr - r["psnr_per_image"][b])
        if b in cmp_["flip_free_images"]:
            assert [len(ys[s][b]) for s in range(20)] == [r["y_len"][s][b] for s in range(20)] and len(zs[b]) == r["z_len"][b]
            assert 8.0 * nbytes / (S * S) == r["bpp_per_image"][b]
            assert [int(out["masks"][i][b].sum().item()) for i in range(10)] == [r["mask_sums"][i][b] for i in range(10)]
            assert d <= NORTH_STAR_PSNR_TOL_DB, f"flip-free image {b}: PSNR differs by {d:.2e} dB"
            worst_ff = max(worst_ff, d)
        else:
            assert d <= 5e-3, f"image {b} (first diverging slice {cmp_['first_diverging_slice'][b]}): PSNR differs by {d:.2e} dB"
            assert abs(8.0 * nbytes / (S * S) - r["bpp_per_image"][b]) <= 5e-3 * r["bpp_per_image"][b]
            worst_fl = max(worst_fl, d)
    bpp = bpp_of(out["strings"], B, S, S)
    psnr = psnr_of(x, x_hat)
    print(f"Config 2 (B=32) vs reference golden: z strings identical {cmp_['z_strings_identical']}/32, y strings {cmp_['y_strings_identical']}/640, "
          f"flip-free images {len(cmp_['flip_free_images'])}/32 (max |dPSNR| {worst_ff:.2e} dB, 1e-4 asserted), flipped images: first diverging slice "
          f"histogram {cmp_['first_diverging_slice_histogram']} (max |dPSNR| {worst_fl:.2e} dB); bpp {bpp:.6f} vs {r['bpp']:.6f}, psnr {psnr:.6f} vs {r['psnr']:.6f}")
    assert abs(bpp - r["bpp"]) <= 1e-3 * r["bpp"] and abs(psnr - r["psnr"]) <= 1e-3
    # floors at the measured values minus a small margin (measured 31 / 32, 10 / 32 flip-free, 31 / 32 within 1e-4 dB) -- the GPU is
    # deterministic, so a change here is a change of the numeric contract or of the kernels' arithmetic, never noise
    within = sum(abs(psnr_of(x[b], x_hat[b]) - r["psnr_per_image"][b]) <= NORTH_STAR_PSNR_TOL_DB for b in range(B))
    assert cmp_["z_strings_identical"] >= 31 and len(cmp_["flip_free_images"]) >= 9 and within >= 28, (cmp_["z_strings_identical"], len(cmp_["flip_free_images"]), within)
    # root flips: the elements that differ from the REFERENCE's inside each image's first diverging slice (tests/golden/config2_roots.npz,
    # the reference's own symbol / index planes) -- the float-rounding flips themselves, asserted as a RATE per coded symbol, and equal to
    # what the contract oracle differs by (the GPU is that oracle bit for bit)
    import bench
    per = 32 * (S // 16) ** 2
    gsym = net.read_tap("sym", np.int32)[: 20 * B * per].reshape(20, B, per)     # planes of the compress() above (decompress leaves them)
    gidx = net.read_tap("idx", np.int32)[: 20 * B * per].reshape(20, B, per)
    rf = bench.root_flips_vs_reference(gsym, gidx, cmp_["first_diverging_slice"])
    print("root flips vs the reference:", {k: v for k, v in rf.items() if k != "per_image_slice_symbols_indexes"})
    assert rf["source"] and not rf["images_whose_first_slice_moved"]
    assert (rf["symbols"], rf["indexes"]) == (rf["expected_by_fixture"]["symbols"], rf["expected_by_fixture"]["indexes"])
    assert rf["rate_per_coded_symbol"] <= 2e-5, rf["rate_per_coded_symbol"]        # ~1e-5 measured: one flip per ~100 000 coded symbols


def test_config3_kodak_sized_set_all_13_levels():
    """BASELINE.json Config 3: the 24 Kodak-sized stand-in images of harness.config3_images -- 18 landscape 512x768,
    6 portrait 768x512 -- through the 13 levels of train.py:293 with compress_with_ac(shared_base=True) (training/step.py:318-365).
    * the whole RD table: finite, bpp non-decreasing over the levels on every image;
    * on image 0 (landscape) and image 3 (portrait): the shared-base strings equal one compress() call per level, string by string,
      and the per-level decode equals the joint decode bit for bit;
    * on the same two images x 13 levels: the REAL reference's fixture tests/golden/config3.json (make_golden_config3.py): shapes,
      hyper-latent strings, and per level either flip-free (all strings identical -> identical bpp, PSNR within 1e-4 dB, identical mask
      sums) or listed with its first diverging slice and held to 5e-3 dB / 2e-3 relative bpp."""
    from progressivecodec_amd.harness import PR_LIST, compare_with_golden_strings, compress_with_ac, config3_images
    imgs = config3_images()
    assert len(imgs) == 24 and sum(1 for x in imgs if x.shape[2] > x.shape[3]) == 6
    net = gpu_codec()
    bpp, psnr, dec_t, rows = compress_with_ac(net, imgs, PR_LIST, shared_base=True)
    assert len(rows) == 24 * 13 and all(math.isfinite(r["bpp"]) and math.isfinite(r["psnr"]) for r in rows)
    for i in range(24):
        b = [rows[i * 13 + l]["bpp"] for l in range(13)]
        assert b == sorted(b), f"image {i}: bpp not monotone over the levels"
    g = _golden_json("config3.json")
    assert g["pr_list"] == PR_LIST
    n_ff = 0
    for gi in g["images"]:
        i = gi["index"]
        x = imgs[i]
        assert (x.shape[2], x.shape[3]) == (gi["H"], gi["W"])
        xc = x.cuda()
        datas = net.compress_levels(xc, PR_LIST, "point-based-std")
        outs = net.decompress_levels([d["strings"] for d in datas], datas[0]["shape"], PR_LIST, "point-based-std")
        for l, (q, lv) in enumerate(zip(PR_LIST, gi["levels"])):
            d = datas[l]
            assert list(d["shape"]) == gi["shape"]
            if l in (0, 4, 7, 12) or i == 3:                      # per-level calls: every level of the portrait image, four of the landscape one
                one = net.compress(xc, q, "point-based-std")
                assert one["strings"] == d["strings"], f"image {i} level {q}: shared-base strings differ from the per-level call"
                one_dec = net.decompress(one["strings"], one["shape"], q, "point-based-std")["x_hat"]
                assert torch.equal(one_dec, outs[l]["x_hat"])
            ys, zs = d["strings"]
            cmp_ = compare_with_golden_strings(d["strings"], [[h] for h in lv["y_sha"]], [lv["z_sha"]])
            assert cmp_["z_strings_identical"] == 1, f"image {i} level {q}: hyper-latent string differs from the reference's"
            x_hat = outs[l]["x_hat"].cpu().clamp_(0, 1)
            p_here = psnr_of(x, x_hat)
            b_here = rows[i * 13 + l]["bpp"]
            assert b_here == bpp_of(d["strings"], 1, gi["H"], gi["W"]) and abs(rows[i * 13 + l]["psnr"] - p_here) < 1e-5   # (the harness reduces the mean on the GPU)
            if cmp_["flip_free_images"]:
                n_ff += 1
                assert b_here == lv["bpp"] and abs(p_here - lv["psnr"]) <= NORTH_STAR_PSNR_TOL_DB
                assert [int(m.sum().item()) for m in d["masks"]] == lv["mask_sums"]
            else:
                # a flipped symbol early in the chain changes the context of everything behind it; at Kodak size and high levels that is
                # up to 3e-3 dB on these untrained synthetic weights (same bound as the Config-2 test's flipped images)
                assert abs(b_here - lv["bpp"]) <= BPP_TOL * max(1.0, lv["bpp"]) and abs(p_here - lv["psnr"]) <= 5e-3
            print(f"Config 3 image {i} ({gi['H']}x{gi['W']}) q={q}: first diverging slice {cmp_['first_diverging_slice'][0]}, bpp {b_here:.6f} (ref {lv['bpp']:.6f}), "
                  f"psnr {p_here:.6f} (ref {lv['psnr']:.6f})")
    print(f"Config 3: {n_ff}/26 (image, level) pairs flip-free against the reference; RD table (24 images) bpp {[round(v, 4) for v in bpp]} psnr {[round(v, 4) for v in psnr]}")
    # Root flips (round 4, tests/golden/config3_roots.npz from make_golden_config3_roots.py): every pair above diverges in a BASE slice, the same
    # for all 13 levels of an image, so the rounding flips that separate the contract from the reference on Config 3 are, per image, the
    # differing elements of that one slice -- counted against the reference's own symbol / index planes; the GPU must reproduce the counts of
    # the contract oracle the fixture was made with, and the rate per coded symbol must stay at the Config-2 level.
    rp = os.path.join(os.path.dirname(__file__), "golden", "config3_roots.npz")
    if os.path.exists(rp):
        R = np.load(rp)
        for n, (i, sl) in enumerate(zip(R["image"].tolist(), R["slice"].tolist())):
            x = imgs[i]
            per = 32 * (x.shape[2] // 16) * (x.shape[3] // 16)
            net.compress(x.cuda(), 0.0, "point-based-std")
            gsym = net.read_tap("sym", np.int32)[: 10 * per].reshape(10, per)
            gidx = net.read_tap("idx", np.int32)[: 10 * per].reshape(10, per)
            ds, di = int((gsym[sl] != R["sym"][n]).sum()), int((gidx[sl] != R["idx"][n]).sum())
            print(f"Config 3 image {i}: root slice {sl} ({per} elements): {ds} symbol and {di} index flips against the reference "
                  f"(fixture: {int(R['contract_sym_flips'][n])} / {int(R['contract_idx_flips'][n])}); {(ds + di) / (10.0 * per):.2e} per coded base symbol")
            assert (ds, di) == (int(R["contract_sym_flips"][n]), int(R["contract_idx_flips"][n]))
            assert (ds + di) / (10.0 * per) <= 5e-5
    # (at Kodak size an (image, level) pair is ~1 M coded symbols: with float-rounding flips at ~1e-5 per symbol a flip-free pair is the
    #  exception -- the count is reported, the tolerances above hold for every pair)


_REM_VARIANT_NETS = {}


def _rem_variant_gpu(c):
    from progressivecodec_amd import ChannelProgresssiveWACNN
    from progressivecodec_amd.rem import PostRateProcessedNetwork
    from progressivecodec_amd.synth import synthetic_post_state_dict
    from tests.util import synth_sd
    key = (c["mu_std"], c["dimension"], c["escalation"])
    if key not in _REM_VARIANT_NETS:
        net = PostRateProcessedNetwork(ChannelProgresssiveWACNN(device="cuda:0"), check_levels=c["check_levels"], mu_std=c["mu_std"],
                                       dimension=c["dimension"], escalation=c["escalation"])
        net.load_state_dict(synth_sd(), synthetic_post_state_dict(3, c["dimension"], mu_std=c["mu_std"]))
        _REM_VARIANT_NETS[key] = net
    return _REM_VARIANT_NETS[key]


@pytest.mark.parametrize("idx", range(6))
def test_rem_variants_bit_exact_vs_oracle_and_reference_goldens(idx):
    """PostRateProcessedNetwork with mu_std=True (the mean refined too), dimension="middle" and escalation / checkpoint_rep
    (CHProgREM.py:15-70, 335-373, 397-416, 773, 989): GPU == contract oracle on every string, mask, the refined mu and scale and x_hat;
    against the REAL reference's fixture (tests/golden/rem_variants.json): hyper-latent strings and shapes identical, flip-free cases
    identical in bytes and within 1e-4 dB."""
    from tests.test_oracle_vs_golden import _rem_variant_cases, rem_variant_oracle, rem_variant_rep
    c = _rem_variant_cases()[idx]
    x = inputs(c["B"], c["H"], c["W"], c["seed"], c["kind"])
    net = _rem_variant_gpu(c)
    orc = rem_variant_oracle(c, "cdet")
    rep_o = rem_variant_rep(orc, c, x)
    rep_g = None
    if c["escalation"]:
        rep_g = net.extract_chekpoint_representation_from_images(x.cuda(), c["checkpoint_quality"])
        assert np.array_equal(rep_g.cpu().numpy().view(np.uint32), rep_o.numpy().view(np.uint32)), "checkpoint representation differs from the oracle's"
    out = net.compress(x.cuda(), c["quality"], "point-based-std", checkpoint_rep=rep_g)
    taps = {}
    orc.set_checkpoint_rep(rep_o)
    ref = orc.compress(x, c["quality"], taps=taps)
    assert out["strings"][1] == ref["strings"][1]
    for s_, (a, b) in enumerate(zip(out["strings"][0], ref["strings"][0])):
        assert a == b, f"y strings of slice {s_} differ"
    for m, rm in zip(out["masks"], ref["masks"]):
        assert np.array_equal(m.cpu().numpy(), rm.numpy())
    for name in ("scale", "mu"):
        t = net.base_net.read_tap(name).reshape(20, c["B"], -1, 32)[13]                      # refined parameters of enhancement slice 3, NHWC
        assert np.array_equal(t.reshape(c["B"], c["H"] // 16, c["W"] // 16, 32).transpose(0, 3, 1, 2), taps["e3"][name].numpy()), name
    dec = net.decompress(out["strings"], out["shape"], c["quality"], "point-based-std", checkpoint_rep=rep_g)
    orc.set_checkpoint_rep(rep_o)
    rdec = orc.decompress(ref["strings"], ref["shape"], c["quality"])["x_hat"]
    assert np.array_equal(dec["x_hat"].cpu().numpy().view(np.uint32), rdec.numpy().view(np.uint32))
    assert [sha(s) for s in out["strings"][1]] == c["z_sha"] and list(out["shape"]) == c["shape"]
    first = flip_report(out["strings"][0], c["y_sha"], c["B"])
    flip_free = all(f is None for f in first)
    psnr = psnr_of(x, dec["x_hat"].cpu().clamp(0, 1))
    print(f"{c['case']} q={c['quality']}: first diverging slice per image {first}; psnr {psnr:.6f} (ref {c['psnr']:.6f})")
    if flip_free:
        assert [[int(m[b].sum().item()) for b in range(c["B"])] for m in out["masks"]] == c["mask_sums"]
    assert abs(psnr - c["psnr"]) <= (NORTH_STAR_PSNR_TOL_DB if flip_free else 5e-2)
    assert abs(bpp_of(out["strings"], c["B"], c["H"], c["W"]) - c["bpp"]) <= (0 if flip_free else 2e-2 * c["bpp"])


def test_harness_batching_same_size_images_gives_the_same_rd_table():
    """compress_with_ac(batch_same_size=True): images of equal size coded in one call per group -- the same RD rows, in input order, as one
    image at a time (an image codes identically alone and inside a batch); on the Config-3 set it is the faster way through the curve."""
    import time
    from progressivecodec_amd.harness import PR_LIST, compress_with_ac, config3_images
    net = gpu_codec()
    imgs = [inputs(1, 64, 128, 41), inputs(1, 96, 72, 42, "smooth"), inputs(1, 64, 128, 43, "smooth"), inputs(1, 96, 72, 44)]
    levels = [PR_LIST[i] for i in (0, 4, 9, 12)]
    b1, p1, _, rows1 = compress_with_ac(net, imgs, levels, shared_base=True)
    b2, p2, _, rows2 = compress_with_ac(net, imgs, levels, batch_same_size=True)
    assert [r["quality"] for r in rows1] == [r["quality"] for r in rows2]
    assert [r["bpp"] for r in rows1] == [r["bpp"] for r in rows2]
    assert max(abs(a["psnr"] - b["psnr"]) for a, b in zip(rows1, rows2)) < 1e-5          # (means reduced on the GPU over different tensors)
    assert b1 == b2
    k3 = config3_images()
    t0 = time.time(); compress_with_ac(net, k3, PR_LIST, shared_base=True); torch.cuda.synchronize(); t1 = time.time()
    compress_with_ac(net, k3, PR_LIST, batch_same_size=True); torch.cuda.synchronize(); t2 = time.time()
    mp = 24 * 512 * 768 * 13 / 1e6
    print(f"Config 3 (24 images x 13 levels, encode + decode): one image at a time {t1 - t0:.2f} s ({mp / (t1 - t0):.1f} level-MP/s), "
          f"same-size images batched {t2 - t1:.2f} s ({mp / (t2 - t1):.1f} level-MP/s)")


# ------------------------------------------------------------------ round 4: the encoder || decoder schedule as a library object
def test_codec_pipeline_equals_sequential_calls():
    """progressivecodec_amd.CodecPipeline.code(): the decode of job i beside the encode of job i+1 on the library's own encoder / decoder
    objects, streams and decoder thread.  Eight different batches (two shapes, three qualities, one multi-level job):
    every string, mask and reconstruction equals what the same calls give one after the other on one object -- and the pipeline built
    AROUND an existing model (from_model) gives the same."""
    from progressivecodec_amd import CodecPipeline
    from tests.util import synth_sd
    net = gpu_codec()
    g = torch.Generator().manual_seed(404)
    jobs = []
    for i in range(8):
        shape = (4, 3, 128, 128) if i % 3 else (2, 3, 64, 192)
        job = {"x": torch.rand(*shape, generator=g).cuda(), "mask_pol": "point-based-std"}
        if i == 5:
            job["qualities"] = [0.0, 0.5, 10.0]
        else:
            job["quality"] = [0.5, 0.0, 2.0][i % 3]
        jobs.append(job)
    want = []
    for job in jobs:                                             # the sequential answer, on one plain model object
        if "qualities" in job:
            enc = net.compress_levels(job["x"], job["qualities"], mask_pol="point-based-std")
            dec = net.decompress_levels([d["strings"] for d in enc], enc[0]["shape"], job["qualities"], mask_pol="point-based-std")
            want.append(([d["strings"] for d in enc], [d["x_hat"].clone() for d in dec], [[m.clone() for m in d["masks"]] for d in enc]))
        else:
            enc = net.compress(job["x"], job["quality"], "point-based-std")
            dec = net.decompress(enc["strings"], enc["shape"], job["quality"], "point-based-std")
            want.append((enc["strings"], dec["x_hat"].clone(), [m.clone() for m in enc["masks"]]))

    def same_x(got, ref, job, enc, what, idx):
        """x_hat of the schedule against the sequential answer; on a mismatch say how they differ and whether a fresh sequential decode of the
        same strings reproduces the reference (i.e. which side moved)"""
        if torch.equal(got, ref):
            return True
        torch.cuda.synchronize()
        d = (got - ref).abs()
        again = net.decompress(enc["strings"], enc["shape"], job["quality"], "point-based-std")["x_hat"] if "quality" in job else None
        per_image = [int((d[b] > 0).sum()) for b in range(d.shape[0])]
        twins = [k for k, w in enumerate(want) if not isinstance(w[1], list) and w[1].shape == got.shape and torch.equal(w[1], got)]
        print(f"[{what}] job {idx} shape {tuple(job['x'].shape)} q={job.get('quality', job.get('qualities'))}: x_hat differs in {int((d > 0).sum())} of {d.numel()} "
              f"elements (per image {per_image}), max |diff| {d.max().item():.3e}, first at {torch.nonzero(d > 0)[0].tolist()}; a fresh sequential decode equals the "
              f"reference: {None if again is None else torch.equal(again, ref)}, equals the pipeline's: {None if again is None else torch.equal(again, got)}; "
              f"the pipeline's x_hat equals the reference of job(s) {twins}")
        return False

    def check(results, what="pipeline"):
        assert len(results) == len(jobs)
        for idx, ((job, enc, dec), ref, j0) in enumerate(zip(results, want, jobs)):
            assert job is j0                                        # in job order
            if "qualities" in job:
                assert [d["strings"] for d in enc] == ref[0]
                assert all(same_x(d["x_hat"], r, job, e, what, idx) for d, r, e in zip(dec, ref[1], enc))
                assert all(torch.equal(m, rm) for d, rms in zip(enc, ref[2]) for m, rm in zip(d["masks"], rms))
            else:
                assert enc["strings"] == ref[0]
                assert same_x(dec["x_hat"], ref[1], job, enc, what, idx)
                assert len(enc["masks"]) == len(ref[2]) and all(torch.equal(m, rm) for m, rm in zip(enc["masks"], ref[2]))

    pipe = CodecPipeline(synth_sd(), device="cuda:0")
    assert pipe.enc._h.value != pipe.dec._h.value and pipe.hw_queues_ok in (True, False)
    check(pipe.run(jobs), "pipe run 1")
    check(pipe.run(jobs), "pipe run 2")                             # the object is reusable
    check(list(pipe.code_sequential(jobs)), "pipe sequential")
    seen = []
    check(list(pipe.code(iter(jobs), on_encoded=lambda job, enc: seen.append(job))), "pipe with callback")
    assert seen == jobs
    pipe2 = CodecPipeline.from_model(net, queue_depth=1)
    assert pipe2.enc is net
    check(pipe2.run(jobs), "from_model, depth 1")
    # a failing job surfaces as its exception on the caller's thread and leaves the pipeline usable
    bad = [jobs[0], {"x": torch.rand(1, 3, 64, 64).cuda(), "quality": 0.5, "mask_pol": "no-such-policy"}, jobs[1]]
    with pytest.raises(NotImplementedError):
        pipe.run(bad)
    check(pipe.run(jobs), "pipe after a failed job")
    # two encoder / decoder pairs: jobs handed to whichever pair is free, results still in job order and identical
    pipe3 = CodecPipeline(synth_sd(), device="cuda:0", n_pairs=2)
    assert len(pipe3.objects) == 4 and len({o._h.value for o in pipe3.objects}) == 4
    check(pipe3.run(jobs), "two pairs run 1")
    check(pipe3.run(jobs * 1), "two pairs run 2")
    with pytest.raises(NotImplementedError):
        pipe3.run(bad)
    check(pipe3.run(jobs), "two pairs after a failed job")
    pr2 = pipe3.profile_conv_in_schedule(jobs)
    assert pr2["jobs"] == len(jobs) and pr2["launches"] > 500
    del pipe3
    # in-schedule profile: brackets only, same results, intervals on one timeline
    pr = pipe.profile_conv_in_schedule(jobs)
    assert pr["jobs"] == len(jobs) and pr["launches"] > 500 and 0 < pr["busy_ms"] <= pr["window_ms"] * 1.001 and pr["sum_ms"] >= pr["busy_ms"] * 0.999
    assert pr["algorithmic_flops"] > 0
    check(pipe.run(jobs), "pipe after profiling")


def test_harness_overlap_gives_the_same_rd_table():
    """compress_with_ac(overlap=True): the reference's loop (training/step.py:297-340) through the library's CodecPipeline -- groups of
    same-size images, the decode of group i beside the encode of group i+1.  Same RD rows as the shared-base loop; and the Config-3 set
    through it (>= 85 level-MP/s asked, 77 without the overlap)."""
    import time
    from progressivecodec_amd.harness import PR_LIST, compress_with_ac, config3_images
    net = gpu_codec()
    imgs = [inputs(1, 64, 128, 41), inputs(1, 96, 72, 42, "smooth"), inputs(1, 64, 128, 43, "smooth"), inputs(1, 96, 72, 44), inputs(1, 64, 128, 45)]
    levels = [PR_LIST[i] for i in (0, 4, 9, 12)]
    b1, p1, _, rows1 = compress_with_ac(net, imgs, levels, shared_base=True)
    b2, p2, t2, rows2 = compress_with_ac(net, imgs, levels, overlap=True, group_size=2)
    assert [r["quality"] for r in rows1] == [r["quality"] for r in rows2]
    assert [r["bpp"] for r in rows1] == [r["bpp"] for r in rows2] and b1 == b2
    assert max(abs(a["psnr"] - b["psnr"]) for a, b in zip(rows1, rows2)) < 1e-5
    assert all(t > 0 for t in t2)
    k3 = config3_images()
    mp = 24 * 512 * 768 * 13 / 1e6
    compress_with_ac(net, k3[:8], PR_LIST, overlap=True)                                   # warm both objects' workspaces
    torch.cuda.synchronize()
    t0 = time.time(); rb = compress_with_ac(net, k3, PR_LIST, batch_same_size=True); torch.cuda.synchronize(); t1 = time.time()
    ro = compress_with_ac(net, k3, PR_LIST, overlap=True); torch.cuda.synchronize(); t2_ = time.time()
    assert rb[0] == ro[0]                                                                    # the same bpp column
    print(f"Config 3 (24 images x 13 levels, encode + decode) through compress_with_ac: batch_same_size {t1 - t0:.2f} s ({mp / (t1 - t0):.1f} level-MP/s), "
          f"overlap=True {t2_ - t1:.2f} s ({mp / (t2_ - t1):.1f} level-MP/s)")


def test_schedule_options_do_not_change_results():
    """pc_codec_set_option: the product's schedule switches (the tuning builds' environment variables are not in this library).  Same strings and
    x_hat under every setting; unknown names and bad values are refused."""
    from progressivecodec_amd import ChannelProgresssiveWACNN
    from progressivecodec_amd._lib import PcodecError
    from tests.util import synth_sd
    x = torch.rand(6, 3, 128, 128, generator=torch.Generator().manual_seed(9)).cuda()
    ref = gpu_codec()
    want = ref.compress(x, 0.5, "point-based-std")
    want_x = ref.decompress(want["strings"], want["shape"], 0.5, "point-based-std")["x_hat"].clone()
    net = ChannelProgresssiveWACNN(device="cuda:0")
    net.load_state_dict(synth_sd())
    for opts in ({"serial_schedule": 1}, {"serial_schedule": 0, "lanes_enc": 2, "lanes_dec": 1}, {"lanes_enc": 3, "lanes_dec": 3}, {"lanes_enc": 0, "lanes_dec": 0, "host_threads": 2},
                 {"host_threads": 0, "profile_in_schedule": 1}, {"profile_in_schedule": 0}):
        for k, v in opts.items():
            net.set_option(k, v)
        out = net.compress(x, 0.5, "point-based-std")
        assert out["strings"] == want["strings"], opts
        assert torch.equal(net.decompress(out["strings"], out["shape"], 0.5, "point-based-std")["x_hat"], want_x), opts
    for name, v in (("no_such_option", 1), ("lanes_enc", 9), ("serial_schedule", -1)):
        with pytest.raises(PcodecError):
            net.set_option(name, v)


def test_reload_of_a_finalised_codec_forgets_the_old_tables():
    """the reference's REM flow -- load the base net, update(), wrap it, load_state_dict(base, post) on the SAME object with a state
    dict that lacks the CDF buffers, update() again.  The reload replaces the native handle; the host-side caches of the old handle's tables
    and scale table must go with it (update() then rebuilds and re-pushes them) -- also when a custom scale table was in force."""
    from progressivecodec_amd import ChannelProgresssiveWACNN
    from progressivecodec_amd.synth import synthetic_state_dict
    sd = synthetic_state_dict()
    assert "gaussian_conditional._quantized_cdf" not in sd or sd["gaussian_conditional._quantized_cdf"].numel() == 0
    x = inputs(2, 64, 64, 21).cuda()
    fresh = ChannelProgresssiveWACNN(device="cuda:0")
    fresh.load_state_dict(sd)
    fresh.update()
    want = fresh.compress(x, 0.5, "point-based-std")["strings"]
    net = ChannelProgresssiveWACNN(device="cuda:0")
    net.load_state_dict(sd)
    assert net.update() is True
    assert net.compress(x, 0.5, "point-based-std")["strings"] == want
    net.load_state_dict(sd)                                        # second load on a finalised object: a fresh native handle
    assert net._gc is None and net._eb is None and net._scale_table is None
    with pytest.raises(ValueError):
        net.compress(x, 0.5, "point-based-std")                    # "Uninitialized CDFs. Run update() first"
    assert net.update() is True
    assert net.compress(x, 0.5, "point-based-std")["strings"] == want
    # custom scale table, then a reload of the plain checkpoint: the native side is back on the checkpoint's table and so are the CDFs
    import math
    custom = torch.exp(torch.linspace(math.log(0.2), math.log(64), 48))
    assert net.update(scale_table=custom, force=True) is True
    other = net.compress(x, 0.5, "point-based-std")["strings"]
    assert other != want
    fresh2 = ChannelProgresssiveWACNN(device="cuda:0")
    fresh2.load_state_dict(sd)
    fresh2.update(scale_table=custom, force=True)
    assert fresh2.compress(x, 0.5, "point-based-std")["strings"] == other
    net.load_state_dict(sd)
    net.update()
    assert net.compress(x, 0.5, "point-based-std")["strings"] == want
    assert net.update(scale_table=custom, force=True) is True      # and the custom table again on the reloaded handle: pushed, not skipped
    assert net.compress(x, 0.5, "point-based-std")["strings"] == other


def test_rem_checkpoint_rep_shape_is_checked_and_never_left_behind():
    """checkpoint_rep must be [B, 320, H/16, W/16] of the call (the native nets index it with the call's own B and h*w); a call
    that fails before the native side consumes the pointer must not leave it for the next one."""
    net = _rem_gpu()
    x = inputs(2, 64, 64, 5, "smooth").cuda()
    want = net.compress(x, 0.5, "point-based-std")
    rep_ok = want["y_hat"]
    assert tuple(rep_ok.shape) == (2, 320, 4, 4)
    with_rep = net.compress(x, 1.0, "point-based-std", checkpoint_rep=rep_ok)["strings"]
    for bad in (rep_ok[:1], rep_ok[:, :, :2], torch.zeros(2, 320, 8, 8), torch.zeros(2, 319, 4, 4), torch.zeros(2, 320, 16)):
        with pytest.raises(ValueError):
            net.compress(x, 1.0, "point-based-std", checkpoint_rep=bad)
        with pytest.raises(ValueError):
            net.decompress(want["strings"], want["shape"], 0.5, "point-based-std", checkpoint_rep=bad)
    # a call that fails inside base_net.compress BEFORE the native call (H not a multiple of 64) with a valid checkpoint_rep ...
    with pytest.raises(ValueError):
        net.compress(torch.rand(2, 3, 72, 64).cuda(), 1.0, "point-based-std", checkpoint_rep=torch.zeros(2, 320, 4, 4))
    # ... leaves nothing behind: the next call without a checkpoint_rep gives the plain answer
    again = net.compress(x, 0.5, "point-based-std")
    assert again["strings"] == want["strings"]
    assert net.compress(x, 1.0, "point-based-std", checkpoint_rep=rep_ok)["strings"] == with_rep


def test_back_to_back_decodes_under_load_keep_their_own_symbols():
    """decompress() returns with work in flight -- x_hat and the chains' last host-to-device symbol copies out of the object's pinned staging.
    The next decompress() of the same object writes that staging from the host at once (the z symbols): it must first wait for the previous
    call's copies.  Round 4 found the missing wait (the CodecPipeline test failed 1 run in 8 under the env matrix: images 0-1 of a batch
    decoded from the NEXT call's z symbols).  Here: a decoder object decodes A then B back to back, many times, while another object keeps
    the GPU queues full (so that A's last copies are still queued when B's host work starts); every x_hat must be the sequential answer."""
    import threading
    from progressivecodec_amd import ChannelProgresssiveWACNN
    from tests.util import synth_sd
    enc = gpu_codec()
    dec = ChannelProgresssiveWACNN(device="cuda:0")
    dec.load_state_dict(synth_sd())
    loads = []
    for _ in range(3):
        o = ChannelProgresssiveWACNN(device="cuda:0")
        o.load_state_dict(synth_sd())
        loads.append(o)
    g = torch.Generator().manual_seed(31)
    xa = torch.rand(4, 3, 128, 128, generator=g).cuda()
    xb = torch.rand(2, 3, 64, 192, generator=g).cuda()
    xl = torch.rand(32, 3, 256, 256, generator=g).cuda()
    a = enc.compress(xa, 0.0, "point-based-std")                  # quality 0: the batch-lane decode path (two lanes, own streams and host threads)
    b = enc.compress(xb, 0.5, "point-based-std")
    a2 = enc.compress(xa, 2.0, "point-based-std")                 # and the base || enhancement pipelined path
    want_a = enc.decompress(a["strings"], a["shape"], 0.0, "point-based-std")["x_hat"].clone()
    want_a2 = enc.decompress(a2["strings"], a2["shape"], 2.0, "point-based-std")["x_hat"].clone()
    want_b = enc.decompress(b["strings"], b["shape"], 0.5, "point-based-std")["x_hat"].clone()
    stop = threading.Event()
    s_dec = torch.cuda.Stream()

    def background(o):
        torch.cuda.set_device(0)
        with torch.cuda.stream(torch.cuda.Stream()):
            while not stop.is_set():
                o.compress(xl, 0.5, "point-based-std")
    ths = [threading.Thread(target=background, args=(o,), daemon=True) for o in loads]
    for t in ths:
        t.start()
    bad = []
    try:
        with torch.cuda.stream(s_dec):
            for it in range(40):
                ra = dec.decompress(a["strings"], a["shape"], 0.0, "point-based-std")["x_hat"]
                rb = dec.decompress(b["strings"], b["shape"], 0.5, "point-based-std")["x_hat"]
                ra2 = dec.decompress(a2["strings"], a2["shape"], 2.0, "point-based-std")["x_hat"]
                rb2 = dec.decompress(b["strings"], b["shape"], 0.5, "point-based-std")["x_hat"]
                s_dec.synchronize()
                for name, got, ref in (("A q=0", ra, want_a), ("B", rb, want_b), ("A q=2", ra2, want_a2), ("B again", rb2, want_b)):
                    if not torch.equal(got, ref):
                        bad.append((it, name, [int((got[i] != ref[i]).sum()) for i in range(got.shape[0])]))
    finally:
        stop.set()
        for t in ths:
            t.join(timeout=120)
    assert not bad, bad[:6]
