"""progressivecodec_amd -- MI355X-native (gfx950) encode/decode hot path of EIDOSLAB/ProgressiveCodec.

Only what ``ChannelProgresssiveWACNN.compress()`` / ``.decompress()`` need lives here:
csrc/ (HIP kernels, C ABI, native runtime -> libpcodec.so), the host mirror of the reference's
model classes (model.py; rem.py for the REM family), the entropy tables / coder surface (entropy.py), the architecture
spec (arch.py), the encoder || decoder schedule (pipeline.py: CodecPipeline) and the synthetic weight generator used by tests and
bench (synth.py).

Importing the package asks the HIP runtime for 16 hardware queues (GPU_MAX_HW_QUEUES, unless the variable is already set or HIP has
already started): the overlapped schedule of CodecPipeline keeps ~20 streams busy and the default 4 queues make its rate bimodal
(pipeline.request_hw_queues).  Nothing else is touched at import; no GPU call is made.
"""
from .arch import CodecConfig, param_spec  # noqa: F401
from .pipeline import request_hw_queues as _request_hw_queues

_request_hw_queues()

__all__ = ["ChannelProgresssiveWACNN", "PostRateProcessedNetwork", "CodecPipeline", "CodecConfig", "param_spec"]


def __getattr__(name):
    if name == "ChannelProgresssiveWACNN":
        from .model import ChannelProgresssiveWACNN
        return ChannelProgresssiveWACNN
    if name == "CodecPipeline":
        from .pipeline import CodecPipeline
        return CodecPipeline
    if name == "PostRateProcessedNetwork":
        from .rem import PostRateProcessedNetwork
        return PostRateProcessedNetwork
    raise AttributeError(name)
