"""ctypes binding of include/s2s_hip.h.  Fails loudly when the HIP library is missing."""
import ctypes as C
import os

from ._build import LIB as _LIB_DEFAULT

_lib = None
loaded_before_torch = False


class S2SConfig(C.Structure):
    _fields_ = [("seq_kmer", C.c_int32), ("max_dna_len", C.c_int32), ("max_signal_len", C.c_int32),
                ("dmodel", C.c_int32), ("dff", C.c_int32), ("n_heads", C.c_int32), ("encoder_layers", C.c_int32),
                ("decoder_layers", C.c_int32), ("pre_layers", C.c_int32), ("scaling_max_value", C.c_float),
                ("compute_mode", C.c_int32)]


class S2SParams(C.Structure):
    _fields_ = [("dwell_mean", C.c_float), ("dwell_std", C.c_float), ("noise_std", C.c_float),
                ("min_noise", C.c_float), ("min_duration", C.c_float), ("noise_sampling", C.c_int32),
                ("duration_sampling", C.c_int32), ("seed", C.c_uint64)]


class S2SDebug(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("emb_out", "enc_out", "sigma", "conc", "rate", "g", "y_scaled", "z01", "emb_in", "dec_in")]


EXPORTS = ("s2s_blob_floats", "s2s_create", "s2s_destroy", "s2s_last_error", "s2s_predict_chunks", "s2s_predict_packed",
           "s2s_export_reads", "s2s_svb_encode", "s2s_philox_u32", "s2s_set_profiling", "s2s_get_kernel_ms", "s2s_stats_read", "s2s_set_attention_path", "s2s_get_attention_path", "s2s_diag_read",
           "s2s_blow5_pack_bound", "s2s_blow5_pack", "s2s_compress_rows", "s2s_sampler_replay", "s2s_sampler_replay_law", "s2s_length_law",
           "s2s_fasta_count", "s2s_fasta_clean", "s2s_fastq_clean", "s2s_copy_ranges", "s2s_blow5_scan", "s2s_blow5_scan_upto", "s2s_attention_redo_threshold")


def lib():
    """The loaded libs2s_hip.so (built by `python -m seq2squiggle_amd._build` / __graft_entry__.build)."""
    global _lib
    if _lib is not None:
        return _lib
    LIB = os.environ.get("S2S_HIP_LIB", _LIB_DEFAULT)   # diagnostic builds are selected explicitly
    if not os.path.exists(LIB):
        raise RuntimeError(f"HIP extension not built: {LIB} is missing (run __graft_entry__.build()); "
                           "there is no CPU fallback for the predict path")
    import sys
    global loaded_before_torch
    loaded_before_torch = "torch" not in sys.modules      # fine for the host-side entry points (the `--gpus N` parent, merge-shards); see engine.py
    L = C.CDLL(LIB)
    vp, i32, i64, u32, u64, f32 = C.c_void_p, C.c_int32, C.c_int64, C.c_uint32, C.c_uint64, C.c_float
    lenient = "S2S_HIP_LIB" in os.environ     # an explicitly selected variant (A/B against an older build): entry points it lacks stay unbound

    def bind(name, restype, argtypes):
        try:
            fn = getattr(L, name)
        except AttributeError:
            if lenient:
                return
            raise
        fn.restype, fn.argtypes = restype, argtypes
    bind("s2s_blob_floats", C.c_size_t, [C.POINTER(S2SConfig)])
    bind("s2s_create", i32, [C.POINTER(S2SConfig), vp, C.c_size_t, i32, C.POINTER(vp)])
    bind("s2s_destroy", None, [vp])
    bind("s2s_last_error", C.c_char_p, [vp])
    bind("s2s_predict_chunks", i32, [vp, vp, vp, vp, i64, i32, C.POINTER(S2SParams), vp, vp, vp, vp, vp,
                                     C.POINTER(S2SDebug)])
    bind("s2s_predict_packed", i32, [vp, vp, vp, vp, vp, i64, i32, C.POINTER(S2SParams), vp, vp])
    bind("s2s_export_reads", i32, [vp, vp, vp, i32, vp, i32, vp, vp, vp, i64, f32, f32, f32, i32])
    bind("s2s_svb_encode", i32, [vp, vp, vp, vp, vp, vp, i32, i64, i32, vp, i64, vp])
    bind("s2s_philox_u32", i32, [vp, vp, u64, u32, u32, u32, u32, i32, vp])
    bind("s2s_set_profiling", i32, [vp, i32])
    bind("s2s_get_kernel_ms", i32, [vp, C.POINTER(C.c_double), C.POINTER(i64), C.POINTER(i64)])
    bind("s2s_stats_read", i32, [vp, C.POINTER(C.c_uint64)])
    bind("s2s_set_attention_path", i32, [vp, i32])
    bind("s2s_get_attention_path", i32, [vp, C.POINTER(i32), C.POINTER(C.c_double)])
    bind("s2s_diag_read", i32, [vp, C.POINTER(C.c_uint64)])
    bind("s2s_blow5_pack_bound", i64, [i64, i32])
    bind("s2s_blow5_pack", i64, [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp, i64])
    bind("s2s_compress_rows", i64, [vp, vp, i32, i32, i32, i32, vp, i64, vp])
    bind("s2s_sampler_replay", i64, [vp, vp, i32, vp, vp, i64, i64, i64, u64, i64, i32, i32, i32, i64, vp, vp])
    bind("s2s_sampler_replay_law", i64, [vp, vp, i32, vp, vp, i64, i64, i64, u64, i64, i32, i32, i32, i64, i32, vp, vp])
    bind("s2s_length_law", i64, [i32, u32, C.c_double, i64])
    bind("s2s_fastq_clean", i64, [vp, i64, i32, vp, vp, vp, i64])
    bind("s2s_fasta_count", i64, [vp, i64])
    bind("s2s_fasta_clean", i64, [vp, i64, i32, vp, vp, vp, i64])
    bind("s2s_copy_ranges", i64, [i32, vp, vp, vp, vp, vp, i32, i32])
    bind("s2s_blow5_scan", i64, [i32, i64, i64])
    bind("s2s_blow5_scan_upto", i64, [i32, i64, i64, i64, C.POINTER(i64)])
    bind("s2s_attention_redo_threshold", C.c_double, [])
    _lib = L
    return L
