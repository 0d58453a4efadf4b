"""seq2squiggle_amd -- MI355X-native engine for the `seq2squiggle predict` hot path.

The device path is hand-written HIP for gfx950 behind a C ABI (include/s2s_hip.h, built into
seq2squiggle_amd/lib/libs2s_hip.so); this package is the Python host side that mirrors the
reference's predict interface (model.py / inference.py of ZKI-PH-ImageAnalysis/seq2squiggle).
There is no CPU fallback: without the HIP library every compute entry point raises.
"""
__version__ = "0.1.0"

from .checkpoint import load_checkpoint, state_dict_to_blob, config_to_c  # noqa: F401
from .chunker import encode_read, encode_reads  # noqa: F401
from .engine import Engine, PredictParams  # noqa: F401
from .modules import Stages  # noqa: F401
