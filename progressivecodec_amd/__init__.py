"""progressivecodec_amd -- MI355X-native (gfx950) encode/decode hot path of EIDOSLAB/ProgressiveCodec.

Only what ``ChannelProgresssiveWACNN.compress()`` / ``.decompress()`` need lives here:
csrc/ (HIP kernels, C ABI, native runtime -> libpcodec.so), the host mirror of the reference's
model classes (model.py; rem.py for the REM family), the entropy tables / coder surface (entropy.py), the architecture
spec (arch.py) and the synthetic weight generator used by tests and bench (synth.py).
"""
from .arch import CodecConfig, param_spec  # noqa: F401

__all__ = ["ChannelProgresssiveWACNN", "PostRateProcessedNetwork", "CodecConfig", "param_spec"]


def __getattr__(name):
    if name == "ChannelProgresssiveWACNN":
        from .model import ChannelProgresssiveWACNN
        return ChannelProgresssiveWACNN
    if name == "PostRateProcessedNetwork":
        from .rem import PostRateProcessedNetwork
        return PostRateProcessedNetwork
    raise AttributeError(name)
