"""ctypes binding of libastts.so (the C ABI declared in include/astts.h).

There is no fallback: if the shared library is missing, or a call returns an error status,
this module raises.  Nothing here computes anything on the CPU.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, c_char_p, c_double, c_float, c_int32, c_int64, c_size_t, c_void_p

# HIP multiplexes its streams onto GPU_MAX_HW_QUEUES hardware queues; streams that share a queue never overlap, and the chip's command
# processor has FOUR pipes: queues that share a pipe take turns at every kernel boundary (two launch chains on such a pair run 2.4x slower
# each, synth/model.py).  FOUR queues = one per pipe: the pipeline's render stream and its (up to three) decode chains then sit on pipes of
# their own by construction and the caller's front stream shares the render queue.  Rounds 1-3 asked for 8 (two queues per pipe): with
# that the three-chain pipeline lost to the two-chain one; with 4 it wins (batch-8 benchmark, same box, alternating: 428 -> 457x real
# time; 5: the same; 3 and 6: no gain; 2: 292x).  Read by the HIP runtime when it initialises (the first HIP call), so it has to be in
# the environment before that; an explicit setting wins.
def _hip_already_initialised() -> bool:
    import sys
    t = sys.modules.get("torch")
    try:
        return bool(t is not None and t.cuda.is_initialized())
    except Exception:      # noqa: BLE001
        return False


if "GPU_MAX_HW_QUEUES" not in os.environ and _hip_already_initialised():
    # too late: the runtime read its own default (8 queues = two per pipe) at its first call.  Everything still works; the stream
    # pipeline then calibrates to two decode chains instead of three.
    import warnings
    warnings.warn("astts: HIP was initialised before astts was imported, so GPU_MAX_HW_QUEUES=4 (one hardware queue per command-processor "
                  "pipe) cannot take effect any more; PipelinedSynth then runs two decode chains instead of three (measured: 428x instead "
                  "of 457x real time on the batch-8 benchmark).  Import astts -- or export GPU_MAX_HW_QUEUES=4 -- before the first "
                  "torch.cuda call.", RuntimeWarning, stacklevel=2)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "4")

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libastts.so")

OK = 0
ERR_INVALID, ERR_HIP, ERR_UNSUPPORTED, ERR_WORKSPACE, ERR_RANGE = -1, -2, -3, -4, -5
DTYPE_F16, DTYPE_F32 = 1, 2
METRIC_COSINE, METRIC_IP, METRIC_L2 = 0, 1, 2
KNN_MAX_K = 1024
KNN_FORCE_EXACT = 1


class AsttsError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"libastts error {code}: {msg}")
        self.code = code


class AsttsLibraryMissing(ImportError):
    pass


_lib = None

# name -> (restype, argtypes); mirrors include/astts.h one to one
_SIGNATURES = {
    "astts_abi_version": (c_int32, []),
    "astts_last_error_string": (c_char_p, []),
    "astts_knn_create": (c_int32, [c_void_p, c_int64, c_int32, c_int32, c_int32, c_void_p, POINTER(c_void_p)]),
    "astts_knn_destroy": (c_int32, [c_void_p]),
    "astts_knn_info": (c_int32, [c_void_p, POINTER(c_int64), POINTER(c_int32), POINTER(c_int32)]),
    "astts_knn_workspace_bytes": (c_size_t, [c_void_p, c_int32, c_int32]),
    "astts_knn_search": (c_int32, [c_void_p, c_void_p, c_int32, c_int32, c_void_p, c_void_p, c_void_p,
                                   c_size_t, c_int32, c_void_p]),
    "astts_knn_search_f64": (c_int32, [c_void_p, c_void_p, c_int32, c_int32, c_void_p, c_void_p, c_void_p, c_void_p,
                                       c_size_t, c_int32, c_void_p]),
    "astts_knn_search_masked": (c_int32, [c_void_p, c_void_p, c_int32, c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p,
                                          c_size_t, c_int32, c_void_p]),
    "astts_knn_last_fallbacks": (c_int32, [c_void_p, c_void_p, c_void_p, POINTER(c_int32)]),
    "astts_knn_profile_enable": (c_int32, [c_void_p, c_int32]),
    "astts_knn_profile_read": (c_int32, [c_void_p, POINTER(c_double), POINTER(c_int64)]),
    # frontend signal processing (astts/audio.py)
    "astts_op_resample_poly": (c_int32, [c_void_p, c_void_p, c_void_p, c_int32, c_int64, c_int64, c_int32, c_int32, c_int32, c_void_p]),
    "astts_op_kaldi_fbank": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int64, c_int32, c_int32, c_int32, c_int32, c_float, c_float,
                                       c_float, c_void_p]),
    "astts_op_whisper_log_mel_workspace_bytes": (c_size_t, [c_int32]),
    "astts_op_whisper_log_mel": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int64, c_int32, c_int32, c_int32, c_void_p, c_size_t,
                                           c_void_p]),
    "astts_op_mel_spectrogram": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int64, c_int32, c_int32, c_int32, c_float,
                                           c_void_p]),
}


def declared_symbols():
    return sorted(_SIGNATURES)


def register_signatures(sigs) -> None:
    """Other host modules (ops) add their part of the ABI here before the first load()."""
    _SIGNATURES.update(sigs)
    if _lib is not None:
        _bind(_lib, sigs)


def _bind(lib, sigs) -> None:
    for name, (res, args) in sigs.items():
        fn = getattr(lib, name)  # AttributeError if the .so does not export it
        fn.restype = res
        fn.argtypes = args


def load():
    """Load libastts.so (once).  Raises AsttsLibraryMissing when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise AsttsLibraryMissing(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            f"or `make -C autostyle-tts_amd/csrc`.  There is no CPU fallback.")
    lib = ctypes.CDLL(LIB_PATH)
    _bind(lib, _SIGNATURES)
    _lib = lib
    return lib


def check(code: int) -> None:
    if code != OK:
        msg = load().astts_last_error_string()
        raise AsttsError(code, msg.decode("utf-8", "replace") if msg else "")


def stream_ptr(stream=None) -> int:
    """hipStream_t of a torch stream (current stream by default) as an integer."""
    import torch

    s = stream if stream is not None else torch.cuda.current_stream()
    return int(s.cuda_stream)
