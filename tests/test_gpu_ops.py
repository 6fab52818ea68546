"""GPU parity of the individual device stages against the CPU oracle (bit-exact).

Every call goes through the C ABI (include/pcodec.h) with raw device pointers.  The oracle
(oracle/pc_oracle.c) evaluates the same fmaf chains / pc_math.h functions on the host.
"""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import liboracle as lo  # noqa: E402  (the checker)


def _lib():
    from progressivecodec_amd._lib import check, lib
    return lib(), check


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def P(t):
    return C.c_void_p(t.data_ptr())


def pack(w, kind):
    L, check = _lib()
    if kind == 0:
        co, ci, k, _ = w.shape
    else:
        ci, co, k, _ = w.shape
    out = np.empty((k * k, ci, co), np.float32)
    check(L.pc_pack_conv_weight(np.ascontiguousarray(w).ctypes.data_as(C.c_void_p), kind, co, ci, k,
                                out.ctypes.data_as(C.c_void_p)))
    return out


def ref_conv(x, w, b, stride, act):
    co, ci, k, _ = w.shape
    pad = k // 2
    B, H, W, _ = x.shape
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    taps = [(ky - pad, kx - pad) for ky in range(k) for kx in range(k)]
    wt = np.ascontiguousarray(w.transpose(2, 3, 1, 0).reshape(k * k, ci, co))
    acc = lo.conv_nhwc(x, wt, taps, stride, Ho, Wo) + b.reshape(1, 1, 1, -1)
    return lo.unary(acc, "gelu") if act else acc


def ref_deconv(x, w, b):
    ci, co, _, _ = w.shape
    B, H, W, _ = x.shape
    out = np.zeros((B, 2 * H, 2 * W, co), np.float32)
    for py in range(2):
        for px in range(2):
            kys, kxs = range(py, 5, 2), range(px, 5, 2)
            taps = [((py + 2 - ky) // 2, (px + 2 - kx) // 2) for ky in kys for kx in kxs]
            wt = np.ascontiguousarray(np.stack([w[:, :, ky, kx] for ky in kys for kx in kxs]))
            lo.conv_nhwc(x, wt, taps, 1, H, W, out=out, ostride=(2, 2), ooff=(py, px))
    return out + b.reshape(1, 1, 1, -1)


CONV_CASES = [
    # B, H, W, Cin, Cout, k, stride, act, tile
    (2, 16, 16, 192, 192, 5, 2, 0, 0),
    (1, 12, 20, 96, 96, 3, 1, 1, 0),
    (2, 8, 8, 352, 224, 3, 1, 1, 0),     # cc stack head, N = 224 (tail of a 64-wide tile)
    (2, 8, 8, 224, 176, 3, 1, 1, 0),     # N = 176 = 5.5 x 32
    (2, 8, 8, 64, 32, 3, 1, 0, 0),       # N = 32
    (3, 4, 4, 288, 256, 3, 2, 1, 0),
    (1, 16, 16, 192, 576, 1, 1, 0, 0),
    (2, 16, 16, 192, 192, 5, 2, 0, 1),   # forced 128x128
    (2, 16, 16, 192, 192, 5, 2, 0, 2),   # forced 64x64
    (2, 16, 16, 192, 192, 5, 2, 0, 3),   # forced 128x32
    (1, 1, 1, 224, 192, 3, 2, 0, 0),     # M = 1 (64x64 image hyper-analysis tail)
    (2, 32, 32, 3, 192, 5, 2, 0, 0),     # first layer: Cin = 3 element-wise gather path
    (8, 64, 64, 96, 96, 3, 1, 1, 0),     # >= 1024 blocks: K-chunk 32 LDS-DMA kernel, N tail 96 = 64 + 32
    (4, 16, 16, 640, 320, 1, 1, 0, 0),   # 1x1 with K = 640 (LDS-DMA kernel, 10 chunks)
    (2, 16, 16, 160, 160, 3, 1, 1, 0),   # Cin = 160: K-chunk 64 with a 32-channel segment tail
    # rows permuted by tap-validity pattern, padding taps skipped per tile (stride 1, images up to 32x32)
    (5, 16, 16, 128, 64, 3, 1, 1, 0),    # 1280 rows: pattern groups end inside tiles
    (3, 4, 4, 192, 96, 3, 1, 0, 0),      # every pixel on the border
    (7, 1, 1, 64, 64, 3, 1, 0, 0),       # only the centre tap is ever valid
    (2, 32, 32, 64, 64, 3, 1, 1, 0),
    (1, 2, 3, 48, 32, 3, 1, 0, 0),
    (2, 16, 16, 32, 32, 5, 1, 0, 0),     # 25 taps
]


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_bit_exact(case):
    L, check = _lib()
    B, H, W, ci, co, k, s, act, tile = case
    rng = np.random.default_rng(hash(case) % (2 ** 32))
    x = rng.standard_normal((B, H, W, ci)).astype(np.float32)
    w = (rng.standard_normal((co, ci, k, k)) * (2.0 / (ci * k * k)) ** 0.5).astype(np.float32)
    b = (rng.standard_normal(co) * 0.1).astype(np.float32)
    want = ref_conv(x, w, b, s, act)
    xd, wd, bd = dev(x), dev(pack(w, 0)), dev(b)
    out = torch.empty(want.shape, device="cuda", dtype=torch.float32)
    check(L.pc_conv2d_nhwc(P(xd), B, H, W, ci, P(wd), P(bd), 0, co, k, s, act, tile, P(out), None))
    torch.cuda.synchronize()
    got = out.cpu().numpy()
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), \
        f"max abs diff {np.abs(got - want).max()} mismatches {(got != want).sum()}/{got.size}"


def _fuzz_conv_cases(n=48, seed=20261005):
    """seeded random layer shapes: row tails (M not a multiple of 32 / 64), column tails (Cout not a multiple of 32), one to three
    channel chunks, all kernel sizes / strides / activations the codec uses, permuted and plain row orders"""
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(n):
        k = int(rng.choice([1, 3, 3, 5]))
        s = int(rng.choice([1, 1, 2])) if k > 1 else 1
        B, H, W = int(rng.integers(1, 4)), int(rng.integers(1, 37)), int(rng.integers(1, 37))
        ci = 16 * int(rng.integers(1, 13))
        co = 4 * int(rng.integers(2, 60))
        out.append((B, H, W, ci, co, k, s, int(rng.integers(0, 2)), 0))
    return out


@pytest.mark.parametrize("case", _fuzz_conv_cases())
def test_conv_bit_exact_random_shapes(case):
    """the unrolled epilogue forms of round 3 (direct / row-table, every tail) against the oracle on random layer shapes"""
    test_conv_bit_exact(case)


@pytest.mark.parametrize("case", [(2, 4, 4, 320, 192), (1, 8, 8, 192, 192), (2, 8, 8, 192, 3)])
def test_deconv_bit_exact(case):
    L, check = _lib()
    B, H, W, ci, co = case
    rng = np.random.default_rng(5 + co)
    x = rng.standard_normal((B, H, W, ci)).astype(np.float32)
    w = (rng.standard_normal((ci, co, 5, 5)) * (8.0 / (ci * 25)) ** 0.5).astype(np.float32)
    b = (rng.standard_normal(co) * 0.1).astype(np.float32)
    want = ref_deconv(x, w, b)
    xd, wd, bd = dev(x), dev(pack(w, 1)), dev(b)
    out = torch.empty(want.shape, device="cuda", dtype=torch.float32)
    check(L.pc_conv2d_nhwc(P(xd), B, H, W, ci, P(wd), P(bd), 1, co, 5, 2, 0, 0, P(out), None))
    torch.cuda.synchronize()
    got = out.cpu().numpy()
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), f"max abs diff {np.abs(got - want).max()}"


def test_deconv_matches_torch_semantics():
    """The phase decomposition is ConvTranspose2d(5, s2, p2, op1) (models/utils.py:196) -- tolerance vs ATen."""
    import torch.nn.functional as F
    rng = np.random.default_rng(9)
    x = rng.standard_normal((1, 6, 5, 32)).astype(np.float32)
    w = (rng.standard_normal((32, 16, 5, 5)) * 0.1).astype(np.float32)
    b = rng.standard_normal(16).astype(np.float32)
    want = F.conv_transpose2d(torch.from_numpy(x).permute(0, 3, 1, 2), torch.from_numpy(w), torch.from_numpy(b),
                              stride=2, padding=2, output_padding=1).permute(0, 2, 3, 1).numpy()
    assert np.allclose(ref_deconv(x, w, b), want, atol=2e-5)


@pytest.mark.parametrize("inverse", [0, 1])
def test_gdn_bit_exact(inverse):
    L, check = _lib()
    rng = np.random.default_rng(3)
    B, H, W, Cc = 2, 9, 7, 192
    x = rng.standard_normal((B, H, W, Cc)).astype(np.float32)
    beta = (1.0 + rng.random(Cc)).astype(np.float32)
    gamma = (0.1 * np.eye(Cc) + 0.004 * np.abs(rng.standard_normal((Cc, Cc)))).astype(np.float32)
    norm = lo.conv_nhwc(x, np.ascontiguousarray(gamma.T).reshape(1, Cc, Cc), [(0, 0)], 1, H, W, square=True) + beta.reshape(1, 1, 1, -1)
    want = x * lo.unary(norm, "sqrt" if inverse else "rsqrt")
    out = torch.empty(want.shape, device="cuda", dtype=torch.float32)
    xd, bd, gd = dev(x), dev(beta), dev(gamma)      # the C ABI takes gamma in the module's [C_out][C_in] layout
    check(L.pc_gdn_nhwc(P(xd), B, H, W, Cc, P(bd), P(gd), inverse, P(out), None))
    torch.cuda.synchronize()
    got = out.cpu().numpy()
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), f"max abs diff {np.abs(got - want).max()}"


@pytest.mark.parametrize("case", [(2, 16, 16, 192, 8, 4), (2, 8, 8, 640, 4, 2), (1, 4, 12, 320, 4, 2), (1, 8, 24, 192, 8, 4)])
def test_window_attention_bit_exact(case):
    L, check = _lib()
    B, H, W, Cc, ws, shift = case
    rng = np.random.default_rng(Cc + ws)
    qkv = rng.standard_normal((B, H, W, 3 * Cc)).astype(np.float32)
    T = ws * ws
    bias = (rng.standard_normal((8, T, T)) * 0.4).astype(np.float32)
    want = lo.win_attention(qkv, bias, 8, ws, shift, np.float32((Cc // 8) ** -0.5))
    out = torch.empty(want.shape, device="cuda", dtype=torch.float32)
    qd, bd = dev(qkv), dev(bias)
    check(L.pc_win_attention_nhwc(P(qd), P(bd), B, H, W, Cc, 8, ws, shift, P(out), None))
    torch.cuda.synchronize()
    got = out.cpu().numpy()
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), f"max abs diff {np.abs(got - want).max()}"


@pytest.mark.parametrize("n_hw,pr", [(16, 0.5), (256, 0.05), (256, 5.0), (1536, 0.75), (4096, 9.99), (64, 2.0), (1024, 1.0), (600, 0.1), (257, 3.0)])
def test_quantile_threshold_bit_exact(n_hw, pr):
    L, check = _lib()
    rng = np.random.default_rng(n_hw)
    B = 3
    scale = (0.6 + 0.7 * rng.standard_normal((B, n_hw, 32))).astype(np.float32)
    scale[1] = np.round(scale[1] * 4) / 4                      # ties
    q = np.float32(1.0 - pr * 0.1)
    want = np.array([lo.quantile(scale[b], q) for b in range(B)], np.float32)
    thr = torch.empty(B, device="cuda", dtype=torch.float32)
    sd = dev(scale)
    check(L.pc_mask_quantile_threshold(P(sd), 32, B, n_hw, 32, q, P(thr), None))
    torch.cuda.synchronize()
    got = thr.cpu().numpy()
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (got, want)
    # and against ATen itself
    t = torch.quantile(torch.from_numpy(scale.reshape(B, -1)), float(1.0 - pr * 0.1), dim=1).numpy()
    assert np.array_equal(got, t)


@pytest.mark.parametrize("kind", ["plain", "q0", "q1", "ties_over_capacity", "constant", "nan", "neg_nan", "unaligned_ld", "ragged_2m", "smooth", "two_clusters"])
def test_quantile_large_image_path(kind):
    """n > 32768: sample -> bracket -> select (pc_stages.hip).  The sample only steers; the result must be the exact order statistic
    whatever the data: heavy ties and adversarial layouts take the verified fallback."""
    L, check = _lib()
    rng = np.random.default_rng(len(kind))
    B, hw, ld, pr = 3, 4096, 32, 0.5
    if kind == "ragged_2m":
        B, hw = 2, 256 * 255 + 3                               # 2.09 M elements: the bracket kernel loops, last step ragged
    if kind == "unaligned_ld":
        ld, hw = 33, 1300                                      # scalar loads
    scale = (0.6 + 0.7 * rng.standard_normal((B, hw, ld))).astype(np.float32)
    if kind == "q0":
        pr = 10.0
    if kind == "q1":
        pr = 0.0
    if kind == "ties_over_capacity":
        scale[0][rng.random((hw, ld)) < 0.6] = 0.11            # 60 % of the image equals the quantile value: more than the list holds
        scale[1] = np.round(scale[1] * 2) / 2
        pr = 6.0
    if kind == "constant":
        scale[:] = 0.25
    if kind == "nan":
        scale[1, 17, 5] = np.nan
    if kind == "neg_nan":
        scale[2, 1000, 31] = np.float32(np.uint32(0xffc00001).view(np.float32))
    if kind == "smooth":                                       # spatially sorted data: runs of 16 consecutive samples are as correlated as can be
        scale = np.sort(scale.reshape(B, -1), axis=1).reshape(B, hw, ld)
        pr = 3.3
    if kind == "two_clusters":                                 # the quantile sits in the empty gap between two clusters
        scale[:, : hw // 2] = 1e-3 * rng.random((B, hw // 2, ld)).astype(np.float32)
        scale[:, hw // 2:] += 100.0
        pr = 5.0
    q = np.float32(1.0 - pr * 0.1)
    want = np.array([lo.quantile(scale[b][:, :32], q) for b in range(B)], np.float32)
    thr = torch.empty(B, device="cuda", dtype=torch.float32)
    sd = dev(scale)
    check(L.pc_mask_quantile_threshold(P(sd), ld, B, hw, 32, q, P(thr), None))
    torch.cuda.synchronize()
    got = thr.cpu().numpy()
    assert np.array_equal(got.view(np.uint32) & 0x7fffffff if "nan" in kind else got.view(np.uint32),
                          want.view(np.uint32) & 0x7fffffff if "nan" in kind else want.view(np.uint32)), (got, want)


@pytest.mark.parametrize("hw,mode,delta", [(16, 0, 0), (256, 1, 1), (100, 1, 1), (256, 2, 1), (1024, 3, 1), (99, 1, 1), (4096 + 36, 1, 0), (61, 0, 1)])
def test_gc_prep_encode_and_decode_bit_exact(hw, mode, delta):
    from tests.util import tables_npz
    L, check = _lib()
    rng = np.random.default_rng(hw + mode)
    B = 2
    table = tables_npz()["scale_table"]
    scale = (0.6 + 0.7 * rng.standard_normal((B, hw, 32))).astype(np.float32)
    scale[0, 0, :4] = table[[3, 10, 20, 62]]                  # exactly on table entries (<= compare)
    mu = rng.standard_normal((B, hw, 32)).astype(np.float32)
    y = (rng.standard_normal((B, hw, 64)) * 2).astype(np.float32)
    y[0, 1, 32:40] = mu[0, 1, :8] + np.array([0.5, 1.5, 2.5, -0.5, -1.5, -2.5, 3.5, 4.5], np.float32)   # ties to even
    thr = np.array([lo.quantile(scale[b], np.float32(0.95)) for b in range(B)], np.float32)
    # oracle
    if mode == 1:
        m = (scale >= thr.reshape(B, 1, 1)).astype(np.float32)
    elif mode == 2:
        m = np.ones_like(scale)
    elif mode == 3:
        m = np.zeros_like(scale)
    else:
        m = None
    ysl = y[..., 32:] - y[..., :32] if delta else y[..., 32:]
    sm = scale if m is None else scale * m
    idx_w = lo.build_indexes(sm, table, 0.11)
    v = ysl - mu
    sym_w = lo.quantize(v if m is None else v * m)
    yhat_w = sym_w.astype(np.float32) + mu
    tr = lambda a: np.ascontiguousarray(a.transpose(0, 2, 1))            # [B][HW][32] -> [B][32][HW]
    sd_, md, yd, td, tabd = dev(scale), dev(mu), dev(y), dev(thr), dev(table)
    sym = torch.empty((B, 32, hw), device="cuda", dtype=torch.int32)
    idx = torch.empty_like(sym)
    msk = torch.empty((B, 32, hw), device="cuda", dtype=torch.float32)
    yhat = torch.zeros((B, hw, 32), device="cuda", dtype=torch.float32)
    ybase = C.c_void_p(yd.data_ptr()) if delta else None
    check(L.pc_gc_prep_encode(P(sd_), 32, P(md), 32, C.c_void_p(yd.data_ptr() + 32 * 4), 64, ybase, 64, P(td), mode, B, hw,
                              P(tabd), 64, 0.11, P(sym), P(idx), P(msk), P(yhat), 32, None))
    idx2 = torch.empty_like(idx)
    check(L.pc_gc_prep_decode_index(P(sd_), 32, P(td), mode, B, hw, P(tabd), 64, 0.11, P(idx2), None, None))
    yhat2 = torch.zeros_like(yhat)
    check(L.pc_gc_dequantize(P(sym), P(md), 32, B, hw, P(yhat2), 32, None))
    torch.cuda.synchronize()
    assert np.array_equal(idx.cpu().numpy(), tr(idx_w))
    assert np.array_equal(idx2.cpu().numpy(), tr(idx_w))
    assert np.array_equal(sym.cpu().numpy(), tr(sym_w))
    if m is not None:
        assert np.array_equal(msk.cpu().numpy(), tr(m))
    assert np.array_equal(yhat.cpu().numpy().view(np.uint32), yhat_w.view(np.uint32))
    assert np.array_equal(yhat2.cpu().numpy().view(np.uint32), yhat_w.view(np.uint32))


def test_quantile_repeated_calls_and_changing_batch():
    """Calls of different batch and image sizes back to back on one stream with no host synchronisation in between (the large-image path
    reuses its scratch: candidate lists, counters): every threshold must still be the exact order statistic."""
    L, check = _lib()
    rng = np.random.default_rng(77)
    cases = [(3, 4096, 0.5), (1, 5000, 2.0), (7, 2048, 9.0), (2, 70000, 0.05), (5, 1500, 5.0), (3, 4096, 0.5), (1, 65281, 1.0), (16, 1100, 3.0), (4, 4096, 7.5)]
    scales = [(0.6 + 0.7 * rng.standard_normal((B, hw, 32))).astype(np.float32) for B, hw, _ in cases]
    scales[2][0] = np.round(scales[2][0] * 8) / 8
    devs = [dev(s) for s in scales]
    outs = [torch.empty(B, device="cuda", dtype=torch.float32) for B, _, _ in cases]
    for rep in range(3):
        for (B, hw, pr), sd, thr in zip(cases, devs, outs):
            check(L.pc_mask_quantile_threshold(P(sd), 32, B, hw, 32, np.float32(1.0 - pr * 0.1), P(thr), None))
    torch.cuda.synchronize()
    for (B, hw, pr), sc, thr in zip(cases, scales, outs):
        want = np.array([lo.quantile(sc[b], np.float32(1.0 - pr * 0.1)) for b in range(B)], np.float32)
        assert np.array_equal(thr.cpu().numpy().view(np.uint32), want.view(np.uint32)), (B, hw, pr)


def test_packed_gelu_equals_the_contract_function_on_every_float():
    """The epilogue's two-elements-per-lane GELU (v_pk_fma_f32 ...; pc_conv.hip: pc_geluf2) against include/pc_math.h: pc_geluf over all
    2^32 arguments: not one differing bit (NaN payloads aside) -- the numeric contract id stays 0x00020001."""
    L, check = _lib()
    bad, nanp = C.c_uint64(), C.c_uint64()
    check(L.pc_selftest_packed_gelu(C.byref(bad), C.byref(nanp)))
    print("packed GELU vs pc_geluf over 2^32 arguments: mismatches", bad.value, "NaN results with another payload", nanp.value)
    assert bad.value == 0
