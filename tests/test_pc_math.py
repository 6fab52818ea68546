"""include/pc_math.h (the float functions of the bitstream contract) against float64 references."""
import numpy as np
from scipy import special

from oracle import liboracle as lo


def ulp_err(val, ref):
    return np.abs(val.astype(np.float64) - ref) / np.spacing(np.abs(ref).astype(np.float32)).astype(np.float64)


def test_accuracy():
    rng = np.random.default_rng(0)
    x = np.concatenate([rng.uniform(-87, 88, 100000), rng.normal(0, 2, 100000), np.linspace(-1, 1, 10001)]).astype(np.float32)
    assert ulp_err(lo.unary(x, "exp"), np.exp(x.astype(np.float64))).max() <= 1.0
    x = np.concatenate([rng.uniform(-5, 5, 200000), rng.normal(0, 1, 100000), np.linspace(-1.01, 1.01, 20001)]).astype(np.float32)
    assert ulp_err(lo.unary(x, "erf"), special.erf(x.astype(np.float64))).max() <= 2.0
    x = np.concatenate([rng.uniform(-10, 10, 200000), rng.normal(0, 0.5, 100000)]).astype(np.float32)
    assert ulp_err(lo.unary(x, "tanh"), np.tanh(x.astype(np.float64))).max() <= 2.0
    assert ulp_err(lo.unary(x, "sigmoid"), special.expit(x.astype(np.float64))).max() <= 2.5
    g = lo.unary(x, "gelu")
    ref = 0.5 * x.astype(np.float64) * (1 + special.erf(x.astype(np.float64) / np.sqrt(2)))
    assert np.abs(g - ref).max() <= 1e-6


def test_special_values_and_rounding():
    e = lo.unary(np.array([-1000, -87.5, 0, 88.6, 100, np.nan], np.float32), "exp")
    assert e[0] == 0 and e[1] == 0 and e[2] == 1 and np.isfinite(e[3]) and np.isinf(e[4]) and np.isnan(e[5])
    assert lo.unary(np.array([-20, 20, 0], np.float32), "tanh").tolist() == [-1.0, 1.0, 0.0]
    assert lo.unary(np.array([-9, 9, 0], np.float32), "erf").tolist() == [-1.0, 1.0, 0.0]
    # torch.round is half-to-even (SURVEY.md section 8c KAT 5)
    assert lo.quantize(np.array([0.5, 1.5, 2.5, -0.5, -1.5, 2.4999, -2.5001], np.float32)).tolist() == [0, 2, 2, 0, -2, 2, -3]


def test_contract_functions_against_the_aten_ops_the_reference_calls():
    """VERDICT r03 "What's weak" 1(b): the GPU and the contract oracle share include/pc_math.h, so GPU == oracle says nothing about how far
    these functions sit from what the REFERENCE evaluates -- torch.erf inside nn.GELU (layers/layers.py:45), torch.tanh (CHProg_cnn.py:761),
    torch.sigmoid (layers.py:73), torch.exp inside softmax (win_attention.py:82), torch.rsqrt (gdn.py:60).  Pinned here against ATen's own
    CPU kernels in float32, in ulps of the ATen result, over dense samples of the ranges the path produces (|x| < 6 for erf / GELU / tanh,
    softmax arguments <= 0, GDN norms in [2^-18, 2^12]).  A flip of a coded symbol needs round(y - mu) to sit within these few ulps of a
    half-integer: the 4.6e-6 per-symbol root-flip rate of tests/golden/config2_roots.npz is what they, and the summation order, leave."""
    import torch
    import torch.nn.functional as F
    rng = np.random.default_rng(5)

    def ulps(ours, theirs):
        t = theirs.astype(np.float64)
        return np.abs(ours.astype(np.float64) - t) / np.maximum(np.spacing(np.abs(theirs)).astype(np.float64), 1e-45)

    x = np.concatenate([rng.uniform(-6, 6, 400000), rng.normal(0, 1, 200000), np.linspace(-1.05, 1.05, 40001), np.linspace(0.99, 4.01, 40001)]).astype(np.float32)
    xt = torch.from_numpy(x)
    worst = {
        "erf": ulps(lo.unary(x, "erf"), torch.erf(xt).numpy()).max(),
        "tanh": ulps(lo.unary(x, "tanh"), torch.tanh(xt).numpy()).max(),
        "sigmoid": ulps(lo.unary(x, "sigmoid"), torch.sigmoid(xt).numpy()).max(),
    }
    xe = np.concatenate([-rng.uniform(0, 30, 300000), -rng.exponential(2.0, 200000), np.zeros(1)]).astype(np.float32)
    worst["exp"] = ulps(lo.unary(xe, "exp"), torch.exp(torch.from_numpy(xe)).numpy()).max()
    g_ours, g_aten = lo.unary(x, "gelu"), F.gelu(xt).numpy()
    worst["gelu_abs"] = float(np.abs(g_ours.astype(np.float64) - g_aten.astype(np.float64)).max())
    print("pc_math.h against ATen CPU float32, worst case:", {k: float(v) for k, v in worst.items()})
    assert worst["erf"] <= 3.0 and worst["tanh"] <= 3.0 and worst["sigmoid"] <= 3.0 and worst["exp"] <= 2.0
    assert worst["gelu_abs"] <= 2e-6                         # measured 1.2e-6 = 2.5 ulps at |GELU| ~ 5 (for x << 0, x * (1 + erf) cancels in both: absolute, not relative)
