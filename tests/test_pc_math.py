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
