"""Import environment for the *real* reference (used only by make_golden.py, in the
build container -- /root/reference does not exist on the GPU box and nothing at
test/run time imports this file's products).

The reference imports two pip packages that are absent here and cannot be
installed (no network):

* ``compressai``  -- only for its native modules ``ans`` / ``_CXX`` (whose sources
  the reference vendors under src/compress/cpp_exts and which oracle/build_ref.sh
  compiles in place), ``available_entropy_coders`` and
  ``compressai.ops.parametrizers.NonNegativeParametrizer`` (the reference carries an
  identical class at src/compress/ops/parametrizers.py:12, re-exported here).
* ``timm.models.layers`` -- ``to_2tuple`` / ``DropPath`` / ``trunc_normal_``; used at
  module-construction time only (win_attention.py:3,81), never in compress()/decompress().

Both are provided as thin import shims generated into oracle/_ref/ (git-ignored).
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF_ROOT = os.environ.get("PC_REFERENCE_ROOT", "/root/reference")
REF_OUT = os.path.join(REPO, "oracle", "_ref")

_SHIMS = {
    "compressai/__init__.py": (
        "from . import ans, _CXX  # built from the reference's vendored sources\n"
        "def available_entropy_coders():\n    return ['ans']\n"
    ),
    "compressai/ops/__init__.py": (
        "def compute_padding(in_h, in_w, *, out_h=None, out_w=None, min_div=1):\n"
        "    if out_h is None: out_h = (in_h + min_div - 1) // min_div * min_div\n"
        "    if out_w is None: out_w = (in_w + min_div - 1) // min_div * min_div\n"
        "    left = (out_w - in_w) // 2; right = out_w - in_w - left\n"
        "    top = (out_h - in_h) // 2; bottom = out_h - in_h - top\n"
        "    return (left, right, top, bottom), (-left, -right, -top, -bottom)\n"
    ),
    "compressai/ops/parametrizers.py": (
        "from compress.ops.parametrizers import NonNegativeParametrizer  # reference's own class\n"
    ),
    "timm/__init__.py": "",
    "timm/models/__init__.py": "",
    "timm/models/layers.py": (
        "import torch\nimport torch.nn as nn\nfrom itertools import repeat\n"
        "def to_2tuple(x):\n    return tuple(x) if isinstance(x, (tuple, list)) else tuple(repeat(x, 2))\n"
        "class DropPath(nn.Identity):\n    def __init__(self, p=0.0):\n        super().__init__()\n"
        "def trunc_normal_(t, mean=0.0, std=1.0, a=-2.0, b=2.0):\n"
        "    return nn.init.trunc_normal_(t, mean=mean, std=std, a=a, b=b)\n"
    ),
}


def setup():
    """Build oracle/_ref (if the reference is present) and put it + the reference on sys.path."""
    if not os.path.isdir(os.path.join(REF_ROOT, "src", "compress")):
        raise RuntimeError(f"reference not present at {REF_ROOT}")
    subprocess.check_call(["bash", os.path.join(REPO, "oracle", "build_ref.sh")])
    for rel, text in _SHIMS.items():
        p = os.path.join(REF_OUT, rel)
        os.makedirs(os.path.dirname(p), exist_ok=True)
        with open(p, "w") as f:
            f.write(text)
    for p in (REF_OUT, os.path.join(REF_ROOT, "src")):
        if p not in sys.path:
            sys.path.insert(0, p)
    sys.dont_write_bytecode = True


def canonical_model():
    """The canonical configuration of SURVEY.md section 8 (wandb-metadata.json:11-20)."""
    setup()
    import torch
    from compress.models import ChannelProgresssiveWACNN
    torch.manual_seed(0)
    net = ChannelProgresssiveWACNN(
        N=192, M=640, division_dimension=[320, 640], dim_chunk=32,
        multiple_decoder=True, multiple_encoder=False, multiple_hyperprior=True,
        mask_policy="two-levels", lmbda_list=[0.0055, 0.04], joiner_policy="res",
        support_progressive_slices=5, delta_encode=True).eval()
    return net
