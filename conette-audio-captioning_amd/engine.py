"""ctypes binding of libconette_hip.so (include/conette_hip.h).

PyTorch is plumbing here: it owns device memory and the stream; every dense stage of the hot
path runs in the hand-written HIP library.  There is NO CPU / eager fallback: if the library
is missing or a call fails, this module raises.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Any, Dict, Optional, Tuple

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libconette_hip.so")

PREC_F32 = 0
PREC_BF16 = 1
PREC_F16X2 = 2   # "exact": fp16 hi/lo operand pairs, three MFMAs per product (include/conette_hip.h)
# (3 was the experimental fp8 precision of ABI 2: withdrawn in round 6 -- e4m3 operands cost the frame embeddings 4 % whatever
#  the scaling, block-scaled MX included: profiles/r06_notes.md section 3)
PREC_F16 = 4     # the bf16 mode's kernels with IEEE fp16 operands: 8x less operand rounding error at the same speed
N_MELS = 224
FEAT = 768
N_TAGS = 527

EXPORTS = (
    "conette_last_error", "conette_abi_version", "conette_create", "conette_destroy", "conette_num_frames",
    "conette_num_audio_frames", "conette_encode_workspace_bytes", "conette_decode_workspace_bytes",
    "conette_frontend_logmel", "conette_encode", "conette_decode", "conette_resample", "conette_resample_len",
    "conette_set_option", "conette_profile_enable", "conette_profile_read", "conette_stream_create_masked",
    "conette_stream_destroy", "conette_forcing_workspace_bytes", "conette_forcing",
    "conette_greedy_workspace_bytes", "conette_greedy", "conette_decode_graph_nodes", "conette_encode_nonfinite",
)
# ---- precision "certified": when is a 16-bit search's decision as good as an exact one's? ------------------------------------
# Per base precision and kind of search, (a, b, c): the top-k call of step i is certified when its margin is at least
# a + b * (i + 1), the final best-beam choice when its margin is at least c.  Calibration (tools/calibrate_margins.py: 512 + 4096
# clips x 2 synthetic checkpoints x greedy / beam 3 against the exact precision with per-call traces, ~10^6 decisions per base
# precision; profiles/r06_margin_calibration*.txt): when a 16-bit search leaves the exact search's trajectory, the margin it saw at
# that call was at most
#     mixed16 0.0156 (greedy) / 0.0218 (beam 3),  f16 0.0312 / 0.0371,  bf16+f16dec 0.097 / 0.184,  bf16 0.193 / 0.290
# -- flat over the 20 steps (the running sums' error is common to the candidates of a parent and cancels in a gap; hence b = 0),
# larger between rows of different parents (beam > 1) than inside one row (greedy); final choice on the averaged scores: at most
# 0.0007 / 0.0014 / 0.013 / 0.019.  The tail is heavy: the 4096-clip maxima are about twice the 512-clip ones (the first
# tolerances, 1.7 x the 512-clip maxima, let ONE greedy clip of a 4096-clip soak through: profiles/r06_certified_soak_first.txt).
# The tolerances below are 2 x the maxima of all 4608 clips per checkpoint.  The certificate is therefore exact in its logic and
# STATISTICAL in its tolerance: tools/certified_soak.py (profiles/r06_certified_soak.txt) counts what gets through on fresh clips;
# what the tolerances cost is the share of clips whose closest decision is nearer than that: `recompute_fraction` of the bench line.
# Other beam sizes (profiles/r06_margin_calibration_beams.txt, base mixed16, 2 048 clips per checkpoint): beams 2 and 5 stay inside the
# beam-3 maxima (0.021 / 0.022; final choice 0.0024), beam 8 reached 0.0385 -- more rows, more chances of a large deviation -- hence a
# third class, "wide" (beam >= 6), at 2 x that; for the other bases the wide tolerances are the beam-3 ones scaled by the same 1.77 and
# were then held to their own runs at beams 2 / 5 / 8 (profiles/r06_margin_calibration_beams_other.txt: f16 reached 0.0405 at beam 2 -- its
# beam tolerance is 2 x that --, bf16 0.024 on the final choice).
# A certificate at beam >= 5 flags 93-100 % of the clips of either synthetic checkpoint anyway.
CERT_TOL = {
    "mixed16": {"greedy": (0.032, 0.0, 0.0), "beam": (0.045, 0.0, 0.005), "wide": (0.08, 0.0, 0.005)},
    "f16": {"greedy": (0.063, 0.0, 0.0), "beam": (0.082, 0.0, 0.008), "wide": (0.13, 0.0, 0.008)},
    "bf16+f16dec": {"greedy": (0.20, 0.0, 0.0), "beam": (0.37, 0.0, 0.026), "wide": (0.65, 0.0, 0.05)},
    "bf16": {"greedy": (0.39, 0.0, 0.0), "beam": (0.58, 0.0, 0.05), "wide": (1.0, 0.0, 0.08)},
}


def cert_kind(beam: int) -> str:
    """the tolerance class of a search: greedy (beam 1), beam (2-5), wide (>= 6)"""
    return "greedy" if int(beam) == 1 else ("beam" if int(beam) <= 5 else "wide")


# The default base: fp16 encoder + EXACT decoder.  Measured against the f16 base in one call (profiles/r06_c_certified_*.json): the exact
# decoder costs every step 0.35 ms, its tighter tolerance spares more exact-ENCODER re-runs than that -- greedy 9.6 k against 8.8 k
# clips/s on the peaked checkpoint (8 % against 16 % of the clips re-run), 6.1 k against 5.1 k on the default one; beam 3 4.8 k against 4.1 k.
CERT_DEFAULT_BASE = "mixed16"
OPT_DECODE_GRAPH = 1
OPT_DECODE_FUSION = 2
OPT_ENCODE_RESERVED_CUS = 3
OPT_FORCING_STEPWISE = 4
MAX_DECODE_GRAPHS = 64   # CONETTE_MAX_DECODE_GRAPHS: decode hipGraphs (= persistent buffer sets) kept per context
PROF_CLASSES = ("frontend", "stem", "dwconv_ln", "pw1_gemm", "pw2_gemm", "downsample", "heads", "dec_prepare",
                "dec_gemm", "dec_attn", "dec_misc", "search")


class ConetteConfigC(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("precision", "vocab_size", "d_model", "nhead", "n_layers", "d_ff", "pad_id",
                                         "bos_id", "eos_id")] + [("reserved", C.c_int32 * 7)]


class EncodeTapsC(C.Structure):
    _fields_ = [("struct_bytes", C.c_size_t), ("logmel", C.c_void_p), ("stem", C.c_void_p), ("stage_block0", C.c_void_p * 4),
                ("stage", C.c_void_p * 4), ("down", C.c_void_p * 4), ("block", C.c_void_p * 18)]


_lib = None


def load_library() -> C.CDLL:
    """dlopen the in-tree library; raises if it has not been built (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(the MI355X path has no CPU fallback)")
    lib = C.CDLL(LIB_PATH)
    lib.conette_last_error.restype = C.c_char_p
    lib.conette_abi_version.restype = C.c_int
    lib.conette_create.restype = C.c_int
    lib.conette_create.argtypes = [C.POINTER(ConetteConfigC), C.c_int32, C.POINTER(C.c_char_p), C.POINTER(C.c_void_p),
                                   C.POINTER(C.c_int64), C.POINTER(C.c_void_p)]
    lib.conette_destroy.restype = None
    lib.conette_destroy.argtypes = [C.c_void_p]
    lib.conette_num_frames.restype = C.c_int32
    lib.conette_num_frames.argtypes = [C.c_int32]
    lib.conette_num_audio_frames.restype = C.c_int32
    lib.conette_num_audio_frames.argtypes = [C.c_int32]
    lib.conette_encode_workspace_bytes.restype = C.c_size_t
    lib.conette_encode_workspace_bytes.argtypes = [C.c_void_p, C.c_int32, C.c_int32]
    lib.conette_decode_workspace_bytes.restype = C.c_size_t
    lib.conette_decode_workspace_bytes.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32]
    lib.conette_frontend_logmel.restype = C.c_int
    lib.conette_frontend_logmel.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]
    lib.conette_encode.restype = C.c_int
    lib.conette_encode.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p,
                                   C.POINTER(EncodeTapsC), C.c_void_p, C.c_size_t, C.c_void_p]
    lib.conette_decode.restype = C.c_int
    lib.conette_decode.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32,
                                   C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                   C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t,
                                   C.c_void_p]
    lib.conette_encode_nonfinite.restype = C.c_int
    lib.conette_encode_nonfinite.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_int32)]
    lib.conette_forcing_workspace_bytes.restype = C.c_size_t
    lib.conette_forcing_workspace_bytes.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32]
    lib.conette_forcing.restype = C.c_int
    lib.conette_forcing.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32,
                                    C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
    lib.conette_greedy_workspace_bytes.restype = C.c_size_t
    lib.conette_greedy_workspace_bytes.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32]
    lib.conette_greedy.restype = C.c_int
    lib.conette_greedy.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32,
                                   C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t,
                                   C.c_void_p]
    lib.conette_set_option.restype = C.c_int
    lib.conette_set_option.argtypes = [C.c_void_p, C.c_int32, C.c_int32]
    lib.conette_decode_graph_nodes.restype = C.c_int32
    lib.conette_decode_graph_nodes.argtypes = [C.c_void_p]
    lib.conette_profile_enable.restype = C.c_int
    lib.conette_profile_enable.argtypes = [C.c_void_p, C.c_uint32]
    lib.conette_profile_read.restype = C.c_int
    lib.conette_profile_read.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_int32)]
    lib.conette_stream_create_masked.restype = C.c_int
    lib.conette_stream_create_masked.argtypes = [C.POINTER(C.c_uint32), C.c_int32, C.POINTER(C.c_void_p)]
    lib.conette_stream_destroy.restype = C.c_int
    lib.conette_stream_destroy.argtypes = [C.c_void_p]
    lib.conette_resample.restype = C.c_int
    lib.conette_resample.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]
    lib.conette_resample_len.restype = C.c_int32
    lib.conette_resample_len.argtypes = [C.c_int32, C.c_int32, C.c_int32]
    if lib.conette_abi_version() != 3:
        raise RuntimeError("libconette_hip.so ABI version mismatch")
    _lib = lib
    return lib


def _check(status: int, what: str) -> None:
    if status != 0:
        raise RuntimeError(f"{what} failed ({status}): {load_library().conette_last_error().decode()}")


def _stream() -> C.c_void_p:
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t: Optional[torch.Tensor]) -> C.c_void_p:
    return C.c_void_p(0 if t is None else t.data_ptr())


def encoder_geometry(n_samples: int) -> Tuple[int, list, list]:
    """(F, H[4], W[4]) -- SURVEY.md A.6."""
    f = n_samples // 320 + 1
    h = [(f + 8 - 4) // 4 + 1]
    w = [56]
    for _ in range(3):
        h.append(h[-1] // 2)
        w.append(w[-1] // 2)
    return f, h, w


def make_partitioned_streams(device: torch.device, decode_share: int = 8, n_cus: int = 256):
    """(encode_stream, decode_stream): the decode stream owns every ``decode_share``-th CU-mask bit
    (1/decode_share of the chip), the encode stream the complement.  Returns torch ExternalStreams."""
    lib = load_library()
    words = (n_cus + 31) // 32
    dec = [0] * words
    for i in range(0, n_cus, decode_share):
        dec[i // 32] |= 1 << (i % 32)
    enc = [(~w) & 0xFFFFFFFF for w in dec]
    out = []
    with torch.cuda.device(device):
        for mask in (enc, dec):
            h = C.c_void_p()
            _check(lib.conette_stream_create_masked((C.c_uint32 * words)(*mask), words, C.byref(h)),
                   "conette_stream_create_masked")
            out.append(torch.cuda.ExternalStream(h.value, device=device))
    return out[0], out[1]


def make_masked_stream(device: torch.device, n_share: int, n_cus: int = 256, offset: int = 0):
    """A stream whose kernels run on ``n_share`` of the ``n_cus`` compute units only: mask bits ``offset .. offset + n_share - 1``.
    On MI355X bit i of the mask is compute unit i // 8 of XCD i % 8 (measured: tools/lab/cumask_probe.hip), so a contiguous run
    of bits takes the same n_share / 8 compute units on every XCD; an XCD whose bits are all clear is left UNRESTRICTED by the
    driver (a mask of every 8th bit therefore restricts nothing).  A decode chain confined this way never holds a compute unit
    one of the encoder's persistent kernels is waiting for, while the encoder's own stream stays unrestricted."""
    lib = load_library()
    words = (n_cus + 31) // 32
    mask = [0] * words
    for j in range(n_share):
        i = (offset + j) % n_cus
        mask[i // 32] |= 1 << (i % 32)
    with torch.cuda.device(device):
        h = C.c_void_p()
        _check(lib.conette_stream_create_masked((C.c_uint32 * words)(*mask), words, C.byref(h)), "conette_stream_create_masked")
        return torch.cuda.ExternalStream(h.value, device=device)


def uncertified_mask(margins: torch.Tensor, best_lprobs: torch.Tensor, tol: Tuple[float, float, float], order: bool = True) -> torch.Tensor:
    """(B,) bool, True = the 16-bit search of this clip is NOT certified to have taken an exact search's decisions: some top-k call of
    step i has a margin below ``a + b * (i + 1)``, or the final best-beam choice one below ``c``, or a score / margin is not finite
    (NaN: fewer finite candidates than picks, or an fp16 residual-stream overflow).  ``margins`` (B, 2, max_pred + 1) as conette_decode
    writes them (include/conette_hip.h): plane 0 membership (+ the final choice in its last column), plane 1 pick order; ``order`` also
    holds the ORDER plane to the tolerance: with it the slot tables (mult_preds / mult_lprobs, in order) are certified, without it
    best_preds / best_lprobs and mult_preds as a SET of hypotheses.  ``tol`` = (a, b, c).  Pure tensor logic (CPU-testable)."""
    a, b, c = tol
    max_pred = margins.shape[2] - 1
    need = a + b * torch.arange(1, max_pred + 1, device=margins.device, dtype=torch.float32)
    ok = (margins[:, 0, :max_pred] >= need).all(dim=1) & (margins[:, 0, max_pred] >= c) & torch.isfinite(best_lprobs)
    if order:
        ok = ok & (margins[:, 1, :max_pred] >= need).all(dim=1)
    return ~ok


def caption_sizes(best_preds: torch.Tensor, mult_preds: torch.Tensor, eos_id: int) -> torch.Tensor:
    """[pred_size, best_maxlen] (beam.py:192-194,207-211,222-225) recomputed from full-width ids -- after rows of two searches have
    been merged.  A hypothesis ends at its first <eos>, or at max_pred when it never emitted one.  Pure tensor logic (CPU-testable)."""
    max_pred = mult_preds.shape[-1]
    pos = torch.arange(max_pred, device=mult_preds.device)

    def first_eos(x):   # index of the first <eos>, max_pred where absent
        return torch.where(x == eos_id, pos, max_pred).amin(dim=-1)

    ps = torch.clamp(first_eos(mult_preds) + 1, max=max_pred).amax()
    e = first_eos(best_preds)
    ml = torch.minimum(torch.where(e < ps, e, ps).amax() + 1, ps)
    return torch.stack([ps, ml]).to(torch.int32)


class Engine:
    """Opaque context (packed weights) + caller-owned workspaces for one device."""

    def __init__(self, state_dict: Dict[str, torch.Tensor], *, precision: str = "bf16", d_model: int = 256,
                 nhead: int = 8, n_layers: int = 6, d_ff: int = 2048, pad_id: int = 0, bos_id: int = 1,
                 eos_id: int = 2, device: Optional[torch.device] = None) -> None:
        if not torch.cuda.is_available():
            raise RuntimeError("conette_amd.Engine needs a ROCm GPU (no CPU fallback)")
        self.lib = load_library()
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        table = {"fp32": PREC_F32, "f32": PREC_F32, "bf16": PREC_BF16, "exact": PREC_F16X2, "f16x2": PREC_F16X2,
                 "f16": PREC_F16, "fp16": PREC_F16}
        # "certified" (round 6): a 16-bit base precision for every clip + the device-side margins of its search (conette_decode's
        # `margins`); clips whose margins do not certify their ids are re-run through an exact context.  "certified" alone picks
        # CERT_DEFAULT_BASE; "certified:<base>" names it; "certified-best[:<base>]" relaxes the slot order (below).
        head = precision.split(":", 1)[0]
        self.certified = head in ("certified", "certified-best")
        # "certified": ids, candidates AND their slot order (mult_preds) are the exact search's; "certified-best": the returned
        # caption (best_preds / best_lprobs) and the SET of beam hypotheses are -- the pick-order margins are not held to the
        # tolerance, so two hypotheses of near-equal score may swap slots; far fewer clips need the exact re-run under beam search
        self.cert_order = head != "certified-best"
        base = precision
        if self.certified:
            base = precision.split(":", 1)[1] if ":" in precision else CERT_DEFAULT_BASE
            if base not in CERT_TOL:
                raise ValueError(f"certified: unknown base precision {base!r} (expected one of {tuple(CERT_TOL)})")
        self.base_precision = base
        if base == "mixed":     # bf16 encoder + exact (fp16 hi/lo pairs) decoder
            self.precision, self.precision_dec = PREC_BF16, PREC_F16X2
        elif base == "mixed16":  # fp16 encoder + exact decoder
            self.precision, self.precision_dec = PREC_F16, PREC_F16X2
        elif base == "bf16+f16dec":  # bf16 encoder (the throughput mode's) + fp16 decoder: the decoder's operands flip the most captions
            self.precision, self.precision_dec = PREC_BF16, PREC_F16
        else:
            if base not in table:
                raise ValueError(f"unknown precision {precision!r}" + (" (the experimental fp8 precision was withdrawn in round 6: "
                                 "profiles/r06_notes.md section 3)" if base == "fp8" else f" (expected one of {sorted(table)}, mixed, mixed16, "
                                 "bf16+f16dec, certified[:base])"))
            self.precision = self.precision_dec = table[base]
        self.precision_name = precision if (self.certified or base in ("mixed", "mixed16", "bf16+f16dec")) else {
            PREC_BF16: "bf16", PREC_F32: "fp32", PREC_F16X2: "exact", PREC_F16: "f16"}[self.precision]
        self.eos_id, self.pad_id = int(eos_id), int(pad_id)
        vocab = int(state_dict["model.decoder.classifier.weight"].shape[0])
        self.vocab_size = vocab
        self.d_model, self.nhead, self.n_layers, self.d_ff = d_model, nhead, n_layers, d_ff
        keep, names, ptrs, numel = [], [], [], []
        with torch.cuda.device(self.device):
            for k, v in state_dict.items():
                if not isinstance(v, torch.Tensor) or k == "_extra_state_":
                    continue
                if v.dtype == torch.bool:
                    v = v.to(torch.uint8)
                elif v.is_floating_point():
                    v = v.to(torch.float32)
                v = v.to(self.device).contiguous()
                keep.append(v)
                names.append(k.encode())
                ptrs.append(v.data_ptr())
                numel.append(v.numel())
            n = len(names)

            def create(prec: int) -> C.c_void_p:
                cfg = ConetteConfigC(prec, vocab, d_model, nhead, n_layers, d_ff, pad_id, bos_id, eos_id)
                handle = C.c_void_p()
                torch.cuda.synchronize(self.device)
                st = self.lib.conette_create(C.byref(cfg), n, (C.c_char_p * n)(*names), (C.c_void_p * n)(*ptrs),
                                             (C.c_int64 * n)(*numel), C.byref(handle))
                _check(st, "conette_create")
                torch.cuda.synchronize(self.device)
                return handle

            self._ctx = create(self.precision)
            # "mixed": frame_embs (B, T, 768) fp32 is the interface between the two halves of the path, so the encoder and the
            # decoder may run at different precisions -- here the bf16 encoder (the throughput mode's) feeds an exact decoder
            self._ctx_dec = create(self.precision_dec) if self.precision_dec != self.precision else self._ctx
            # "certified": the exact context that re-runs the clips the margins do not certify (encoder + decoder; shared with
            # the base's decoder when that is exact already)
            self._ctx_x = None
            if self.certified:
                self._ctx_x = self._ctx_dec if self.precision_dec == PREC_F16X2 else create(PREC_F16X2)
        self.cert_stats = {"clips": 0, "recomputed": 0}
        self._ws: Dict[str, torch.Tensor] = {}
        self._dec_bufs: Dict[Any, Dict[str, Any]] = {}
        del keep

    def __del__(self) -> None:
        seen = []
        for c in (getattr(self, "_ctx", None), getattr(self, "_ctx_dec", None), getattr(self, "_ctx_x", None)):
            if c and all(c is not d for d in seen):
                seen.append(c)
                try:
                    self.lib.conette_destroy(c)
                except Exception:
                    pass
        self._ctx = self._ctx_dec = self._ctx_x = None

    def _workspace(self, key: str, nbytes: int) -> torch.Tensor:
        ws = self._ws.get(key)
        if ws is None or ws.numel() < nbytes:
            ws = torch.empty(int(nbytes), dtype=torch.uint8, device=self.device)
            self._ws[key] = ws
        return ws

    # ---- a2 --------------------------------------------------------------------------------
    def frontend_logmel(self, wave: torch.Tensor) -> torch.Tensor:
        wave = wave.to(self.device, torch.float32).contiguous()
        b, l = wave.shape
        out = torch.empty((b, self.lib.conette_num_frames(l), N_MELS), dtype=torch.float32, device=self.device)
        _check(self.lib.conette_frontend_logmel(self._ctx, _ptr(wave), b, l, _ptr(out), _stream()), "frontend_logmel")
        return out

    # ---- a2-a7 -----------------------------------------------------------------------------
    def encode(self, wave: torch.Tensor, taps=False, out: Optional[Tuple[torch.Tensor, torch.Tensor]] = None,
               slot: int = 0, exact: bool = False):
        """wave (B, L) fp32 on device -> frame_embs (B, T, 768), clip_probs (B, 527) [, taps dict].
        ``taps="blocks"`` additionally returns the output of every ConvNeXt block ("block0" .. "block17").
        ``slot`` selects the scratch workspace: encodes running concurrently on different streams need different slots.
        ``exact`` (certified engines only): run on the exact context that re-computes uncertified clips."""
        ctx = self._ctx
        if exact:
            if self._ctx_x is None:
                raise RuntimeError("encode(exact=True) needs a certified engine")
            ctx = self._ctx_x
        wave = wave.to(self.device, torch.float32).contiguous()
        b, l = wave.shape
        f, hs, ws_ = encoder_geometry(l)
        t = hs[3]
        if out is None:
            frame_embs = torch.empty((b, t, FEAT), dtype=torch.float32, device=self.device)
            clip = torch.empty((b, N_TAGS), dtype=torch.float32, device=self.device)
        else:
            frame_embs, clip = out
        need = self.lib.conette_encode_workspace_bytes(ctx, b, l)
        wsb = self._workspace(("xenc" if exact else "enc") + ("" if slot == 0 else str(slot)), need)
        tap_struct, tap_out = None, None
        if taps:
            dims = (96, 192, 384, 768)
            e = lambda *s: torch.empty(s, dtype=torch.float32, device=self.device)
            tap_out = {"logmel": e(b, f, N_MELS), "stem": e(b, hs[0], ws_[0], 96)}
            for i in range(4):
                tap_out[f"stage{i}_block0"] = e(b, hs[i], ws_[i], dims[i])
                tap_out[f"stage{i}"] = e(b, hs[i], ws_[i], dims[i])
                if i > 0:
                    tap_out[f"down{i}"] = e(b, hs[i], ws_[i], dims[i])
            tap_struct = EncodeTapsC()
            tap_struct.struct_bytes = C.sizeof(EncodeTapsC)
            tap_struct.logmel = tap_out["logmel"].data_ptr()
            tap_struct.stem = tap_out["stem"].data_ptr()
            for i in range(4):
                tap_struct.stage_block0[i] = tap_out[f"stage{i}_block0"].data_ptr()
                tap_struct.stage[i] = tap_out[f"stage{i}"].data_ptr()
                tap_struct.down[i] = tap_out[f"down{i}"].data_ptr() if i > 0 else 0
            if taps == "blocks":
                blk = 0
                for i, depth in enumerate((3, 3, 9, 3)):
                    for _ in range(depth):
                        tap_out[f"block{blk}"] = e(b, hs[i], ws_[i], dims[i])
                        tap_struct.block[blk] = tap_out[f"block{blk}"].data_ptr()
                        blk += 1
        st = self.lib.conette_encode(ctx, _ptr(wave), b, l, _ptr(frame_embs), _ptr(clip),
                                     C.byref(tap_struct) if tap_struct is not None else None, _ptr(wsb),
                                     wsb.numel(), _stream())
        _check(st, "conette_encode")
        if taps:
            return frame_embs, clip, tap_out
        return frame_embs, clip

    # ---- a9-a14 ----------------------------------------------------------------------------
    def _decode_buffers(self, b: int, t: int, beam: int, max_pred: int, s0: bool, trace: bool,
                        slot: int = 0, margins: bool = False, exact: bool = False) -> Dict[str, Any]:
        """Persistent I/O buffers per shape (and pipeline slot): identical pointers let the library
        replay its hipGraph."""
        key = (b, t, beam, max_pred, s0, trace, slot, margins, exact)
        buf = self._dec_bufs.pop(key, None)
        if buf is not None:
            self._dec_bufs[key] = buf           # least recently used first: a hit moves to the back
        if buf is None:
            dev = self.device
            ldv = (self.vocab_size + 7) // 8 * 8
            e = lambda shape, dt: torch.empty(shape, dtype=dt, device=dev)
            buf = {
                "fe": e((b, t, FEAT), torch.float32), "lens": e((b,), torch.int32), "bos": e((b,), torch.int32),
                "forbid": e((self.vocab_size,), torch.uint8),
                "best_preds": e((b, max_pred), torch.int32), "best_lprobs": e((b,), torch.float32),
                "mult_preds": e((b, beam, max_pred), torch.int32), "mult_lprobs": e((b, beam), torch.float32),
                "sizes": e((2,), torch.int32),
                "step0": e((b * beam, ldv), torch.float32) if s0 else None,
                "trace_sel": e((max_pred, b, beam, 2), torch.int32) if trace else None,
                "trace_val": e((max_pred, b, beam), torch.float32) if trace else None,
                "margins": e((b, 2, max_pred + 1), torch.float32) if margins else None,
            }
            if len(self._dec_bufs) >= MAX_DECODE_GRAPHS:
                # the evicted buffers may still be written by a decode running on another stream: drain before they are freed
                torch.cuda.synchronize(self.device)
                self._dec_bufs.pop(next(iter(self._dec_bufs)))
            self._dec_bufs[key] = buf
        return buf

    def decode(self, frame_embs: torch.Tensor, frame_lens: torch.Tensor, bos_ids: torch.Tensor,
               forbid_mask: Optional[torch.Tensor], beam: int, min_pred: int, max_pred: int,
               want_step0_logits: bool = False, want_trace: bool = False, clone: bool = True,
               slot: int = 0, want_margins: bool = False, exact: bool = False) -> Dict[str, torch.Tensor]:
        """Beam search over pre-computed frame embeddings.  Outputs are full width; trim with
        ``sizes`` = [pred_size, best_maxlen].  ``clone=False`` returns the persistent buffers of
        pipeline ``slot`` (two slots let the decode of batch i overlap the encode of batch i+1).
        ``want_margins``: also ``margins`` (B, 2, max_pred + 1), the per-decision margins of the search -- plane 0 membership (+ the
        final choice in its last column), plane 1 pick order (include/conette_hip.h).
        ``exact`` (certified engines only): run on the exact context."""
        ctx = self._ctx_dec
        if exact:
            if self._ctx_x is None:
                raise RuntimeError("decode(exact=True) needs a certified engine")
            ctx = self._ctx_x
        b, t, _ = frame_embs.shape
        buf = self._decode_buffers(b, t, int(beam), int(max_pred), want_step0_logits, want_trace, slot, want_margins, exact)
        if frame_embs.data_ptr() != buf["fe"].data_ptr():
            buf["fe"].copy_(frame_embs, non_blocking=True)
        buf["lens"].copy_(frame_lens.to(torch.int32), non_blocking=True)
        buf["bos"].copy_(bos_ids.to(torch.int32), non_blocking=True)
        forbid_ptr = None
        if forbid_mask is not None:
            if forbid_mask.numel() != self.vocab_size:
                raise ValueError("forbid_mask must have vocab_size entries")
            buf["forbid"].copy_(forbid_mask.to(torch.uint8), non_blocking=True)
            forbid_ptr = buf["forbid"]
        need = self.lib.conette_decode_workspace_bytes(ctx, b, t, beam, max_pred)
        # per slot: decodes of different slots may run on different streams
        wsb = self._workspace(("xdec" if exact else "dec") + ("" if slot == 0 else str(slot)), need)
        st = self.lib.conette_decode(ctx, _ptr(buf["fe"]), _ptr(buf["lens"]), _ptr(buf["bos"]), _ptr(forbid_ptr),
                                     b, t, beam, min_pred, max_pred, _ptr(buf["best_preds"]), _ptr(buf["best_lprobs"]),
                                     _ptr(buf["mult_preds"]), _ptr(buf["mult_lprobs"]), _ptr(buf["sizes"]),
                                     _ptr(buf["step0"]), _ptr(buf["trace_sel"]), _ptr(buf["trace_val"]), _ptr(buf["margins"]),
                                     _ptr(wsb), wsb.numel(), _stream())
        _check(st, "conette_decode")
        cp = (lambda x: x.clone()) if clone else (lambda x: x)
        out = {k: cp(buf[k]) for k in ("best_preds", "best_lprobs", "mult_preds", "mult_lprobs", "sizes")}
        if want_step0_logits:
            out["step0_logits"] = cp(buf["step0"])[:, : self.vocab_size]
        if want_trace:
            out["trace_sel"] = cp(buf["trace_sel"])
            out["trace_val"] = cp(buf["trace_val"])
        if want_margins:
            out["margins"] = cp(buf["margins"])
        return out

    # ---- the id certificate (round 6) ------------------------------------------------------------------------------------
    def uncertified(self, margins: torch.Tensor, best_lprobs: torch.Tensor, tol: Optional[Tuple[float, float, float]] = None,
                    beam: int = 2, order: Optional[bool] = None) -> torch.Tensor:
        """``uncertified_mask`` with the engine's defaults: ``tol`` = ``CERT_TOL[base precision][cert_kind(beam)]`` (measured:
        tools/calibrate_margins.py, profiles/r06_margin_calibration*.txt), ``order`` = the engine's policy (True for "certified",
        False for "certified-best")."""
        tol = CERT_TOL[self.base_precision][cert_kind(beam)] if tol is None else tol
        return uncertified_mask(margins, best_lprobs, tol, self.cert_order if order is None else order)

    def caption_sizes(self, best_preds: torch.Tensor, mult_preds: torch.Tensor) -> torch.Tensor:
        return caption_sizes(best_preds, mult_preds, self.eos_id)

    def generate_certified(self, wave: Optional[torch.Tensor], frame_embs: torch.Tensor, frame_lens: torch.Tensor,
                           bos_ids: torch.Tensor, forbid_mask: Optional[torch.Tensor], beam: int, min_pred: int,
                           max_pred: int, tol: Optional[Tuple[float, float, float]] = None) -> Dict[str, torch.Tensor]:
        """The certified search: the base precision's search of every clip with margins; the clips it does not certify are
        compacted into one batch and re-run through the exact context -- from the waveform (``wave`` (B, L), the padded batch
        the embeddings came from: exact encoder + exact decoder) or, when ``wave`` is None (the caller gave embeddings:
        ``preprocess=False``, BaselinePLM), from the same embeddings (exact decoder) -- and scattered back.  One host
        round trip per BATCH (the number of such clips sizes the second launch), none per search step.
        Returns decode()'s dict + ``recomputed`` (B,) bool."""
        if not self.certified:
            raise RuntimeError("generate_certified needs Engine(precision='certified[:base]')")
        res = self.decode(frame_embs, frame_lens, bos_ids, forbid_mask, beam, min_pred, max_pred, want_margins=True)
        flag = self.uncertified(res["margins"], res["best_lprobs"], tol, beam)
        if wave is None and self.precision_dec == PREC_F16X2:
            flag = torch.zeros_like(flag)   # given embeddings + an exact decoder already: nothing a re-run could change
        idx = torch.nonzero(flag).flatten()
        n = int(idx.numel())            # (the host round trip)
        b = int(frame_embs.shape[0])
        self.cert_stats["clips"] += b
        self.cert_stats["recomputed"] += n
        res["recomputed"] = flag
        if n == 0:
            return res
        lens_d = frame_lens.to(self.device)
        bos_d = bos_ids.to(self.device)
        if wave is not None:
            fe_x, clip_x = self.encode(wave.index_select(0, idx), exact=True)
            res["recomputed_idx"], res["recomputed_clip_probs"] = idx, clip_x    # (the exact encoder's tag probabilities of those clips)
        else:
            fe_x = frame_embs.to(self.device).index_select(0, idx)
        rx = self.decode(fe_x, lens_d.index_select(0, idx), bos_d.index_select(0, idx), forbid_mask, beam, min_pred, max_pred,
                         exact=True)
        for k in ("best_preds", "best_lprobs", "mult_preds", "mult_lprobs"):
            res[k].index_copy_(0, idx, rx[k])
        res["sizes"] = self.caption_sizes(res["best_preds"], res["mult_preds"])
        return res

    def encode_nonfinite(self) -> int:
        """Positions whose LayerNorm statistics were not finite in the encodes of this process since the last call (the fp16
        residual stream of the 16-bit precisions overflowed: include/conette_hip.h); waits for the current stream."""
        n = C.c_int32(0)
        _check(self.lib.conette_encode_nonfinite(self._ctx, _stream(), C.byref(n)), "conette_encode_nonfinite")
        return int(n.value)

    def forcing(self, frame_embs: torch.Tensor, frame_lens: torch.Tensor, caps_in: torch.Tensor) -> torch.Tensor:
        """Teacher forcing (forcing.py:12-71): frame_embs (B, T, 768), caps_in (B, cap_len) ids right-padded with
        pad_id -> logits (B, cap_len, vocab) fp32."""
        b, t, _ = frame_embs.shape
        cap_len = int(caps_in.shape[1])
        fe = frame_embs.to(self.device, torch.float32).contiguous()
        lens = frame_lens.to(self.device, torch.int32).contiguous()
        caps = caps_in.to(self.device, torch.int32).contiguous()
        out = torch.empty((b, cap_len, self.vocab_size), dtype=torch.float32, device=self.device)
        need = self.lib.conette_forcing_workspace_bytes(self._ctx_dec, b, t, cap_len)
        wsb = self._workspace("dec", need)
        st = self.lib.conette_forcing(self._ctx_dec, _ptr(fe), _ptr(lens), _ptr(caps), b, t, cap_len, _ptr(out), _ptr(wsb),
                                      wsb.numel(), _stream())
        _check(st, "conette_forcing")
        return out

    def greedy(self, frame_embs: torch.Tensor, frame_lens: torch.Tensor, bos_ids: torch.Tensor,
               forbid_mask: Optional[torch.Tensor], min_pred: int, max_pred: int) -> Dict[str, torch.Tensor]:
        """greedy_search (greedy.py:17-131): {"logits": (B, pred_size, vocab) masked step logits, "preds": (B, pred_size)}."""
        b, t, _ = frame_embs.shape
        fe = frame_embs.to(self.device, torch.float32).contiguous()
        lens = frame_lens.to(self.device, torch.int32).contiguous()
        bos = bos_ids.to(self.device, torch.int32).contiguous()
        fm = None if forbid_mask is None else forbid_mask.to(self.device, torch.uint8).contiguous()
        logits = torch.empty((b, max_pred, self.vocab_size), dtype=torch.float32, device=self.device)
        preds = torch.empty((b, max_pred), dtype=torch.int32, device=self.device)
        sizes = torch.zeros((2,), dtype=torch.int32, device=self.device)
        need = self.lib.conette_greedy_workspace_bytes(self._ctx_dec, b, t, max_pred)
        wsb = self._workspace("dec", need)
        st = self.lib.conette_greedy(self._ctx_dec, _ptr(fe), _ptr(lens), _ptr(bos), _ptr(fm), b, t, int(min_pred), int(max_pred),
                                     _ptr(logits), _ptr(preds), _ptr(sizes), _ptr(wsb), wsb.numel(), _stream())
        _check(st, "conette_greedy")
        ps = int(sizes[0].item())
        return {"logits": logits[:, :ps].contiguous(), "preds": preds[:, :ps].contiguous()}

    def decode_input_buffer(self, b: int, t: int, beam: int, max_pred: int, slot: int = 0, margins: bool = False,
                            exact: bool = False) -> torch.Tensor:
        """The persistent (B, T, 768) input of decode() (of the call with the same ``want_margins`` / ``exact``): encode
        straight into it to skip a copy."""
        return self._decode_buffers(b, t, int(beam), int(max_pred), False, False, slot, margins, exact)["fe"]

    # ---- options / profiling ------------------------------------------------------------------
    def set_decode_graph(self, enabled: bool) -> None:
        _check(self.lib.conette_set_option(self._ctx_dec, OPT_DECODE_GRAPH, int(bool(enabled))), "set_option")

    def set_decode_fusion(self, enabled: bool) -> None:
        """bf16: fused decoder-layer kernels (default) or one launch per sub-layer (cross-check path)."""
        _check(self.lib.conette_set_option(self._ctx_dec, OPT_DECODE_FUSION, int(bool(enabled))), "set_option")

    def set_forcing_stepwise(self, enabled: bool) -> None:
        """Teacher forcing through the KV-cached step kernels (cross-check) instead of the one-pass kernels (default)."""
        _check(self.lib.conette_set_option(self._ctx_dec, OPT_FORCING_STEPWISE, int(bool(enabled))), "set_option")

    def set_encode_reserved_cus(self, n: int) -> None:
        """Compute units the encoder's persistent kernels leave free for a decode running on another stream."""
        _check(self.lib.conette_set_option(self._ctx, OPT_ENCODE_RESERVED_CUS, int(n)), "set_option")

    def decode_graph_nodes(self) -> int:
        """Kernel / copy nodes of the most recently captured decode hipGraph (0 before the first capture)."""
        return int(self.lib.conette_decode_graph_nodes(self._ctx_dec))

    _DEC_CLASSES = ("dec_prepare", "dec_gemm", "dec_attn", "dec_misc", "search")

    def profile_enable(self, classes=()) -> None:
        """Event pairs around the launches of these kernel classes.  In the two-context precisions (mixed, mixed16,
        bf16+f16dec) the encoder classes go to the encoder context and the decoder classes to the decoder context (which also
        makes its decode run eagerly instead of replaying a hipGraph, as in the one-context case)."""
        mask_enc = mask_dec = 0
        for c in classes:
            bit = 1 << PROF_CLASSES.index(c)
            if c in self._DEC_CLASSES:
                mask_dec |= bit
            else:
                mask_enc |= bit
        if self._ctx_dec is self._ctx:
            _check(self.lib.conette_profile_enable(self._ctx, mask_enc | mask_dec), "profile_enable")
        else:
            _check(self.lib.conette_profile_enable(self._ctx, mask_enc), "profile_enable")
            _check(self.lib.conette_profile_enable(self._ctx_dec, mask_dec), "profile_enable")

    def profile_read(self) -> Dict[str, Any]:
        n = len(PROF_CLASSES)
        ms = (C.c_float * n)()
        cnt = (C.c_int32 * n)()
        _check(self.lib.conette_profile_read(self._ctx, ms, cnt), "profile_read")   # (accumulates into ms / cnt)
        if self._ctx_dec is not self._ctx:
            _check(self.lib.conette_profile_read(self._ctx_dec, ms, cnt), "profile_read")
        return {PROF_CLASSES[i]: (float(ms[i]), int(cnt[i])) for i in range(n) if cnt[i] > 0}

    # ---- a1 --------------------------------------------------------------------------------
    def resample(self, x: torch.Tensor, orig_sr: int, new_sr: int) -> torch.Tensor:
        """(..., n) fp32 -> (..., ceil(n * new / orig)); torchaudio sinc_interpolation defaults."""
        if orig_sr == new_sr:
            return x
        shape = x.shape
        x2 = x.to(self.device, torch.float32).reshape(-1, shape[-1]).contiguous()
        n_out = self.lib.conette_resample_len(shape[-1], int(orig_sr), int(new_sr))
        out = torch.empty((x2.shape[0], n_out), dtype=torch.float32, device=self.device)
        _check(self.lib.conette_resample(_ptr(x2), x2.shape[0], shape[-1], int(orig_sr), int(new_sr), _ptr(out),
                                         _stream()), "conette_resample")
        return out.reshape(*shape[:-1], n_out)
