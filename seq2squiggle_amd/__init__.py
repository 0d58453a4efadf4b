"""seq2squiggle_amd -- MI355X-native engine for the `seq2squiggle predict` hot path.

The device path is hand-written HIP for gfx950 behind a C ABI (include/s2s_hip.h, built into
seq2squiggle_amd/lib/libs2s_hip.so); this package is the Python host side that mirrors the
reference's predict interface (model.py / inference.py of ZKI-PH-ImageAnalysis/seq2squiggle).
There is no CPU fallback: without the HIP library every compute entry point raises.
"""
__version__ = "0.1.0"

# The public names resolve on first use (PEP 562): `python -m seq2squiggle_amd predict ... --gpus N` and `merge-shards` run in a
# parent process that must start in milliseconds and must not load torch (its ranks do, as children) -- importing the package
# eagerly cost that parent a second of the command's wall clock.
_EXPORTS = {"load_checkpoint": "checkpoint", "state_dict_to_blob": "checkpoint", "config_to_c": "checkpoint",
            "encode_read": "chunker", "encode_reads": "chunker", "Engine": "engine", "PredictParams": "engine", "Stages": "modules"}
__all__ = sorted(_EXPORTS) + ["__version__"]


def __getattr__(name):
    if name in _EXPORTS:
        import importlib
        value = getattr(importlib.import_module(f".{_EXPORTS[name]}", __name__), name)
        globals()[name] = value
        return value
    try:                                     # sub-modules as attributes (seq2squiggle_amd.signal_io, ...), as after an eager import
        import importlib
        return importlib.import_module(f".{name}", __name__)
    except ImportError:
        raise AttributeError(f"module {__name__!r} has no attribute {name!r}") from None


def __dir__():
    return __all__
