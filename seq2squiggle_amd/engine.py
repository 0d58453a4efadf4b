"""Engine: the Python face of the C ABI (include/s2s_hip.h).

PyTorch is used for device memory and streams only: tensors are allocated with torch on the
engine's device and handed to the library as raw pointers; launches go to torch's current
stream so they order with the caller's other work.
"""
import ctypes as C
import os
from dataclasses import dataclass
from typing import Dict, Optional

import numpy as np
import torch

from . import _lib
from .checkpoint import config_to_c, load_checkpoint, state_dict_to_blob

T_ENC, T_DEC = 16, 250


@dataclass
class PredictParams:
    """The scalars predict_step reads from the reference LightningModule (model.py:55-63)."""
    dwell_mean: float = 12.5
    dwell_std: float = 0.0
    noise_std: float = 2.0
    noise_sampling: bool = True
    duration_sampling: bool = True
    min_noise: float = 0.0
    min_duration: float = 3.0
    seed: int = 0

    def to_c(self) -> "_lib.S2SParams":
        return _lib.S2SParams(float(self.dwell_mean), float(self.dwell_std), float(self.noise_std),
                              float(self.min_noise), float(self.min_duration), int(bool(self.noise_sampling)),
                              int(bool(self.duration_sampling)), int(self.seed) & 0xFFFFFFFFFFFFFFFF)


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


class Engine:
    """Weights resident on one GPU + the predict / export entry points."""

    def __init__(self, state_dict: Dict[str, torch.Tensor], config: dict, device: Optional[int] = None,
                 mode: str = "f16x3"):
        self._h = None
        L = _lib.lib()                       # raises when the HIP extension is missing
        if not torch.cuda.is_available():
            raise RuntimeError("seq2squiggle_amd needs a ROCm GPU (gfx950); there is no CPU fallback")
        self.device_index = torch.cuda.current_device() if device is None else int(device)
        self.device = torch.device("cuda", self.device_index)
        self.config = dict(config)
        self._pe_dec_host = state_dict["decoders.position_enc"].detach().float().reshape(1, T_DEC, 64).clone()
        self._pe_dec = None
        self.k = int(config["seq_kmer"])
        self.mode = mode
        ccfg = config_to_c(config, mode)
        blob = np.ascontiguousarray(state_dict_to_blob(state_dict, config))
        h = C.c_void_p()
        with torch.cuda.device(self.device):
            torch.cuda.current_stream().synchronize()
            rc = L.s2s_create(C.byref(ccfg), blob.ctypes.data_as(C.c_void_p), blob.nbytes, self.device_index, C.byref(h))
        if rc != 0:
            hint = ""
            if rc == -2 and _lib.loaded_before_torch:      # S2S_ERR_HIP with the library bound to another HIP runtime than torch's
                hint = (" -- libs2s_hip.so was loaded BEFORE torch in this process (it then binds to /opt/rocm's libamdhip64 instead of the "
                        "one torch bundles, and sees no device): import torch first")
            raise ValueError(f"s2s_create failed ({rc}): {L.s2s_last_error(None).decode()}{hint}")
        self._h = h
        if os.environ.get("S2S_PROFILE_KERNEL"):          # diagnostic: HIP events around every predict launch, summed up at close()
            self.set_profiling(True)

    @classmethod
    def from_checkpoint(cls, path: str, device: Optional[int] = None, mode: str = "f16x3") -> "Engine":
        sd, cfg = load_checkpoint(path)
        return cls(sd, cfg, device, mode)

    def decoder_position_enc(self) -> torch.Tensor:
        """decoders.position_enc [1,250,64] on the engine's device (the stand-alone Decoder operator adds it, modules.py:136)."""
        if self._pe_dec is None:
            self._pe_dec = self._pe_dec_host.to(self.device)
        return self._pe_dec

    def close(self):
        if self._h is not None and os.environ.get("S2S_PROFILE_KERNEL"):
            import sys
            ms, nl, nc = self.kernel_ms()
            print(f"[S2S_PROFILE_KERNEL] {nl} launches, {nc} chunks, {ms:.1f} ms of predict kernel", file=sys.stderr)
        if self._h is not None:
            _lib.lib().s2s_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc: int, what: str):
        if rc != 0:
            raise RuntimeError(f"{what} failed ({rc}): {_lib.lib().s2s_last_error(self._h).decode()}")

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    # ------------------------------------------------------------------ predict
    def predict_chunks(self, bases: torch.Tensor, n_valid: torch.Tensor, params: PredictParams,
                       first_global_chunk: int = 0, inject_g: Optional[torch.Tensor] = None,
                       inject_zdw: Optional[torch.Tensor] = None, inject_z01: Optional[torch.Tensor] = None,
                       debug: bool = False, out_signal: Optional[torch.Tensor] = None,
                       out_dur: Optional[torch.Tensor] = None, emb_in: Optional[torch.Tensor] = None,
                       dec_in: Optional[torch.Tensor] = None):
        """bases uint8 [B, 16+k-1] and n_valid uint8 [B] on the engine's device ->
        dict(signal fp32 [B,250] pA, dur int32 [B,16] [, debug stage tensors]).
        emb_in [B,16,64] / dec_in [B,250,64] (float32): stage inputs taken from these tensors instead of being computed from the
        bases -- the stand-alone sub-module operators of seq2squiggle_amd.modules (s2s_debug.emb_in / dec_in)."""
        B = int(bases.shape[0])
        nb = T_ENC + self.k - 1
        if bases.dtype != torch.uint8 or bases.dim() != 2 or bases.shape[1] != nb or not bases.is_contiguous():
            raise ValueError(f"bases must be contiguous uint8 [B, {nb}]")
        if n_valid.dtype != torch.uint8 or n_valid.shape != (B,) or not n_valid.is_contiguous():
            raise ValueError("n_valid must be contiguous uint8 [B]")
        for name, t, shape in (("inject_g", inject_g, (B, T_ENC)), ("inject_zdw", inject_zdw, (B, T_ENC)),
                               ("inject_z01", inject_z01, (B, T_DEC)), ("emb_in", emb_in, (B, T_ENC, 64)),
                               ("dec_in", dec_in, (B, T_DEC, 64))):
            if t is not None and (t.dtype != torch.float32 or tuple(t.shape) != shape or not t.is_contiguous()
                                  or t.device != self.device):
                raise ValueError(f"{name} must be contiguous float32 {shape} on {self.device}")
        if bases.device != self.device or n_valid.device != self.device:
            raise ValueError(f"inputs must live on {self.device}")
        for name, t, dt, shape in (("out_signal", out_signal, torch.float32, (B, T_DEC)),
                                   ("out_dur", out_dur, torch.int32, (B, T_ENC))):
            if t is not None and (t.dtype != dt or tuple(t.shape) != shape or not t.is_contiguous() or t.device != self.device):
                raise ValueError(f"{name} must be contiguous {dt} {shape} on {self.device}")
        sig = out_signal if out_signal is not None else torch.empty(B, T_DEC, dtype=torch.float32, device=self.device)
        dur = out_dur if out_dur is not None else torch.empty(B, T_ENC, dtype=torch.int32, device=self.device)
        out = {"signal": sig, "dur": dur}
        dbg = None
        if debug:
            f = dict(dtype=torch.float32, device=self.device)
            out.update(emb_out=torch.zeros(B, 16, 64, **f), enc_out=torch.zeros(B, 16, 64, **f),
                       sigma=torch.zeros(B, 16, **f), conc=torch.zeros(B, 16, **f), rate=torch.zeros(B, 16, **f),
                       g=torch.zeros(B, 16, **f), y_scaled=torch.zeros(B, T_DEC, **f), z01=torch.zeros(B, T_DEC, **f))
            dbg = _lib.S2SDebug(*[out[n].data_ptr() for n in ("emb_out", "enc_out", "sigma", "conc", "rate", "g",
                                                              "y_scaled", "z01")])
        if emb_in is not None or dec_in is not None:
            dbg = dbg or _lib.S2SDebug()
            dbg.emb_in = None if emb_in is None else emb_in.data_ptr()
            dbg.dec_in = None if dec_in is None else dec_in.data_ptr()
        p = params.to_c()
        with torch.cuda.device(self.device):
            rc = _lib.lib().s2s_predict_chunks(self._h, self._stream(), _ptr(bases), _ptr(n_valid),
                                               int(first_global_chunk), B, C.byref(p), _ptr(inject_g), _ptr(inject_zdw),
                                               _ptr(inject_z01), _ptr(sig), _ptr(dur), C.byref(dbg) if dbg else None)
        self._check(rc, "s2s_predict_chunks")
        return out

    def predict_packed(self, read_bytes: torch.Tensor, chunk_start: torch.Tensor, n_valid: torch.Tensor,
                       params: PredictParams, first_global_chunk: int = 0):
        """Chunks addressed inside a packed read buffer (chunker.pack_reads): read_bytes uint8 [N], chunk_start int64
        [B], n_valid uint8 [B], all on the engine's device -> dict(signal [B,250], dur [B,16])."""
        B = int(chunk_start.shape[0])
        for name, t, dt in (("read_bytes", read_bytes, torch.uint8), ("chunk_start", chunk_start, torch.int64),
                            ("n_valid", n_valid, torch.uint8)):
            if t.dtype != dt or not t.is_contiguous() or t.device != self.device or t.dim() != 1:
                raise ValueError(f"{name} must be a contiguous 1-D {dt} tensor on {self.device}")
        if n_valid.shape[0] != B:
            raise ValueError("n_valid and chunk_start differ in length")
        sig = torch.empty(B, T_DEC, dtype=torch.float32, device=self.device)
        dur = torch.empty(B, T_ENC, dtype=torch.int32, device=self.device)
        p = params.to_c()
        with torch.cuda.device(self.device):
            rc = _lib.lib().s2s_predict_packed(self._h, self._stream(), _ptr(read_bytes), _ptr(chunk_start), _ptr(n_valid),
                                               int(first_global_chunk), B, C.byref(p), _ptr(sig), _ptr(dur))
        self._check(rc, "s2s_predict_packed")
        return {"signal": sig, "dur": dur}

    # ------------------------------------------------------------------ export
    def export_reads(self, signal: torch.Tensor, read_first: torch.Tensor, digitisation: float = 0.0,
                     signal_range: float = 1.0, offset_mean: float = 0.0, rna: bool = False, want_pa: bool = True,
                     want_dac: bool = False, out_offsets: torch.Tensor = None, out_dac: torch.Tensor = None):
        """Per-read zero-strip (model.py:284-286) and optional int16 conversion (signal_io.py:134-141) on
        the GPU.  signal [B,250]; read_first int32 [R+1] -> dict(offsets int64 [R+1], pa, dac).  out_offsets / out_dac: write
        there instead of into fresh tensors (int64 [R+1] / int16 [>= B*250], e.g. two views of one buffer that then leaves the
        device as a single copy)."""
        B, R = int(signal.shape[0]), int(read_first.shape[0]) - 1
        if signal.dtype != torch.float32 or not signal.is_contiguous() or signal.shape[1] != T_DEC:
            raise ValueError("signal must be contiguous float32 [B,250]")
        if read_first.dtype != torch.int32 or not read_first.is_contiguous() or read_first.dim() != 1 or R < 0:
            raise ValueError("read_first must be contiguous int32 [R+1]")
        if signal.device != self.device or read_first.device != self.device:
            raise ValueError(f"signal and read_first must live on {self.device}")
        cap = B * T_DEC
        for name, t, dt, n in (("out_offsets", out_offsets, torch.int64, R + 1), ("out_dac", out_dac, torch.int16, cap)):
            if t is not None and (t.dtype != dt or t.dim() != 1 or not t.is_contiguous() or t.device != self.device or t.numel() < n):
                raise ValueError(f"{name} must be a contiguous 1-D {dt} tensor of at least {n} elements on {self.device}")
        offs = out_offsets[:R + 1] if out_offsets is not None else torch.empty(R + 1, dtype=torch.int64, device=self.device)
        pa = torch.empty(cap, dtype=torch.float32, device=self.device) if want_pa else None
        dac = None
        if want_dac or out_dac is not None:
            dac = out_dac[:cap] if out_dac is not None else torch.empty(cap, dtype=torch.int16, device=self.device)
        with torch.cuda.device(self.device):
            rc = _lib.lib().s2s_export_reads(self._h, self._stream(), _ptr(signal), B, _ptr(read_first), R, _ptr(offs),
                                             _ptr(pa), _ptr(dac), cap, float(digitisation), float(signal_range),
                                             float(offset_mean), int(bool(rna)))
        self._check(rc, "s2s_export_reads")
        return {"offsets": offs, "pa": pa, "dac": dac}

    @staticmethod
    def svb_capacity(total_samples_bound: int, n_rows: int, variant: int) -> int:
        """Bytes s2s_svb_encode may write for n_rows rows holding at most total_samples_bound samples together."""
        if variant == 32:      # per row: u32 count + ceil(n/4) control bytes + up to 3 bytes per zig-zag delta of int16 samples
            return 4 * n_rows + 3 * total_samples_bound + (total_samples_bound + 3 * n_rows) // 4
        return 2 * total_samples_bound + total_samples_bound // 8 + n_rows

    def svb_encode(self, dac: torch.Tensor, read_offsets: torch.Tensor, row_read: torch.Tensor, row_index: torch.Tensor,
                   row_samples: int, variant: int, total_samples_bound: int, out: torch.Tensor = None,
                   out_offsets: torch.Tensor = None):
        """StreamVByte blobs of the rows (see s2s_svb_encode in include/s2s_hip.h) -> dict(out uint8 [capacity],
        offsets int64 [N+1]) on the device; offsets[N] < 0 reports a too-small `out` (minus the bytes needed; rows that did
        not fit are unwritten).  total_samples_bound: an upper bound of the samples the rows hold together.
        out / out_offsets: write there (uint8 [>= svb_capacity(...)] / int64 [N+1]) instead of into fresh tensors."""
        N = int(row_read.shape[0])
        for name, t, dt in (("dac", dac, torch.int16), ("read_offsets", read_offsets, torch.int64),
                            ("row_read", row_read, torch.int32), ("row_index", row_index, torch.int32)):
            if t.dtype != dt or t.dim() != 1 or not t.is_contiguous() or t.device != self.device:
                raise ValueError(f"{name} must be a contiguous 1-D {dt} tensor on {self.device}")
        if row_index.shape[0] != N or variant not in (16, 32):
            raise ValueError("row_read / row_index differ in length, or variant is not 16 | 32")
        cap = self.svb_capacity(total_samples_bound, N, variant)
        for name, t, dt, n in (("out", out, torch.uint8, cap), ("out_offsets", out_offsets, torch.int64, N + 1)):
            if t is not None and (t.dtype != dt or t.dim() != 1 or not t.is_contiguous() or t.device != self.device or t.numel() < n):
                raise ValueError(f"{name} must be a contiguous 1-D {dt} tensor of at least {n} elements on {self.device}")
        out = out if out is not None else torch.empty(max(cap, 1), dtype=torch.uint8, device=self.device)
        offs = out_offsets[:N + 1] if out_offsets is not None else torch.empty(N + 1, dtype=torch.int64, device=self.device)
        with torch.cuda.device(self.device):
            rc = _lib.lib().s2s_svb_encode(self._h, self._stream(), _ptr(dac), _ptr(read_offsets), _ptr(row_read),
                                           _ptr(row_index), N, int(row_samples), int(variant), _ptr(out), cap, _ptr(offs))
        self._check(rc, "s2s_svb_encode")
        return {"out": out, "offsets": offs}

    # ------------------------------------------------------------------ misc
    def philox_u32(self, seed: int, c0: int, c1: int, c2: int, c3: int, n: int) -> torch.Tensor:
        out = torch.empty(n, 4, dtype=torch.int32, device=self.device)
        with torch.cuda.device(self.device):
            rc = _lib.lib().s2s_philox_u32(self._h, self._stream(), int(seed), c0, c1, c2, c3, n, _ptr(out))
        self._check(rc, "s2s_philox_u32")
        return out

    def set_profiling(self, enabled: bool):
        self._check(_lib.lib().s2s_set_profiling(self._h, int(enabled)), "s2s_set_profiling")

    def kernel_ms(self):
        """-> (total predict-kernel ms, launches, chunks) since the last call (HIP events on the launch stream)."""
        ms, nl, nc = C.c_double(), C.c_int64(), C.c_int64()
        self._check(_lib.lib().s2s_get_kernel_ms(self._h, C.byref(ms), C.byref(nl), C.byref(nc)), "s2s_get_kernel_ms")
        return ms.value, nl.value, nc.value

    @property
    def attention_path(self) -> str:
        """"fast" (fast softmax path first, exact path on overflow) or "exact": chosen per checkpoint by s2s_create's calibration
        launch, or set here (s2s_set_attention_path)."""
        p, r = C.c_int32(), C.c_double()
        self._check(_lib.lib().s2s_get_attention_path(self._h, C.byref(p), C.byref(r)), "s2s_get_attention_path")
        return "exact" if p.value else "fast"

    @attention_path.setter
    def attention_path(self, path: str):
        if path not in ("fast", "exact"):
            raise ValueError("attention_path must be 'fast' or 'exact'")
        self._check(_lib.lib().s2s_set_attention_path(self._h, int(path == "exact")), "s2s_set_attention_path")

    @property
    def calibration_redo_rate(self) -> float:
        p, r = C.c_int32(), C.c_double()
        self._check(_lib.lib().s2s_get_attention_path(self._h, C.byref(p), C.byref(r)), "s2s_get_attention_path")
        return r.value

    def stats(self) -> dict:
        """Counters of the predict kernel since the last call (s2s_stats_read; synchronises the device): how THIS run behaved --
        the share of softmax runs redone on the safe path and the clock the SIMDs held are data dependent."""
        out = (C.c_uint64 * 10)()
        self._check(_lib.lib().s2s_stats_read(self._h, out), "s2s_stats_read")
        chunks, runs, redo, cyc, ticks, wgs, exact_chunks = (int(x) for x in out[:7])
        return {"chunks": chunks, "softmax_runs": runs, "softmax_redone": redo,
                "redo_rate": (redo / runs) if runs else 0.0,
                "in_kernel_clock_ghz": (cyc / ticks * 0.1) if ticks else None,
                "cycles_per_chunk_and_cu": (cyc / chunks) if chunks else None,    # a workgroup owns its CU: sum of their cycles / chunks
                "shader_cycles": cyc, "ticks_100mhz": ticks, "workgroups": wgs,
                "chunks_on_exact_path": exact_chunks}
