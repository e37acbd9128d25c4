"""ctypes binding of libcbgx.so (include/cbgx.h).  There is no fallback: if the library is missing
or a call fails, this raises."""
import contextlib
import contextvars
import ctypes
import os

import torch  # noqa: F401  (loads the HIP runtime libcbgx links against, by SONAME)

from .build import LIBPATH, XCHECK_LIBPATH

_LIB = None
_XLIB = None

_vp, _i, _sz = ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t
ABI_VERSION = 6      # include/cbgx.h CBGX_ABI_VERSION: bumped whenever the packed-weight layout or an entry point changes

EXPORTS = {
    "cbgx_abi_version": (_i, []),
    "cbgx_last_error": (ctypes.c_char_p, []),
    "cbgx_set_edge_workgroups": (_i, [_i]),
    "cbgx_packed_weights_floats": (_sz, [_i, _i]),
    "cbgx_pack_weights": (_i, [ctypes.POINTER(_vp), _i, _i, _i, _vp, _vp]),
    "cbgx_workspace_bytes": (_sz, [_i, _i]),
    "cbgx_unitransformer_forward": (_i, [_vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _vp, _vp, _vp, _sz, _vp]),
    "cbgx_unitransformer_forward_cached": (_i, [_vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp,
                                                _vp, _vp, _vp, ctypes.c_uint, _vp, _sz, _vp]),
    "cbgx_knn_graph": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp, _vp]),
    "cbgx_edge_gate": (_i, [_vp, _vp, _vp, _vp, _i, _vp, _vp]),
    "cbgx_x2h_attention": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _sz, _vp]),
    "cbgx_h2x_attention": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _sz, _vp]),
    "cbgx_classifier": (_i, [_vp, _i, _i, _vp, _i, _vp, _vp, _sz, _vp]),
    "cbgx_packed_h2x_stack_floats": (_sz, [_i]),
    "cbgx_pack_h2x_stack": (_i, [ctypes.POINTER(_vp), _i, _i, _vp, _vp]),
    "cbgx_h2x_stack_forward": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _vp, _sz, _vp]),
    "cbgx_targetdiff_prologue": (_i, [_vp, _vp, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "cbgx_targetdiff_epilogue": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, ctypes.POINTER(_vp), _vp, _vp, _vp, _vp,
                                      _vp, _vp]),
    "cbgx_targetdiff_step_boundary": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, ctypes.POINTER(_vp), _vp, _vp, _vp, _vp,
                                           _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "cbgx_targetdiff_prologue_traj": (_i, [_vp, _vp, _vp, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "cbgx_targetdiff_epilogue_traj": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp, ctypes.POINTER(_vp), _vp, _vp, _vp]),
    "cbgx_diffbp_epilogue": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _i, _vp, _vp, _vp,
                                  _vp, _vp]),
    "cbgx_diffsbdd_step": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, ctypes.c_float, ctypes.c_float,
                                ctypes.c_float, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "cbgx_targetdiff_train_noise": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "cbgx_targetdiff_loss": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, ctypes.POINTER(_vp), _vp, _vp, _vp,
                                  _vp, _vp, _vp]),
    "cbgx_targetdiff_loss_backward": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    "cbgx_diffsbdd_train_noise": (_i, [_vp] * 9 + [_i] * 4 + [_vp, _vp, _i] + [_vp] * 5),
    "cbgx_diffsbdd_loss": (_i, [_vp] * 7 + [_i] * 4 + [_vp] * 8),
    "cbgx_compose_plan": (_i, [_vp, _vp, _i, _i, _i] + [_vp] * 7),
    "cbgx_embed_compose": (_i, [_vp] * 8 + [_i] * 5 + [ctypes.POINTER(_vp)] + [_vp] * 5),
    "cbgx_embed_compose_backward": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _i, _vp, _vp]),
    "cbgx_diffbp_loss": (_i, [_vp] * 13 + [_i, _i, _i, _i, _vp, _vp, ctypes.c_float, ctypes.c_float] + [_vp] * 9 + [_vp]),
    "cbgx_train_tape_bytes": (_sz, [_i, _i]),
    "cbgx_train_workspace_bytes": (_sz, [_i]),
    "cbgx_unitransformer_forward_train": (_i, [_vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _vp, _vp, _vp, _sz,
                                               _vp, _sz, _vp]),
    "cbgx_unitransformer_forward_train_ex": (_i, [_vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _vp, _vp, ctypes.c_uint, _vp, _sz,
                                                  _vp, _sz, _vp]),
    "cbgx_unitransformer_backward": (_i, [_vp, _i, _i, _vp, _sz, _vp, _vp, _i, _vp, _vp, _vp, ctypes.POINTER(_vp), _i,
                                          _vp, _vp, _sz, _vp]),
    "cbgx_x2h_attention_backward": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp,
                                         ctypes.POINTER(_vp), _vp, _sz, _vp]),
    "cbgx_h2x_attention_backward": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp,
                                         ctypes.POINTER(_vp), _vp, _sz, _vp]),
    "cbgx_h2x_stack_tape_bytes": (_sz, [_i, _i]),
    "cbgx_h2x_stack_forward_train": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _vp, _sz, _vp, _sz, _vp]),
    "cbgx_h2x_stack_backward": (_i, [_vp, _i, _vp, _sz, _vp, _vp, _vp, _i, _vp, ctypes.POINTER(_vp), _i, _vp, _vp, _sz,
                                     _vp]),
    "cbgx_profile_begin": (_i, [_i]),
    "cbgx_profile_end": (_i, [ctypes.POINTER(ctypes.c_double), ctypes.POINTER(_i), _i]),
}

PROFILE_CLASSES = ("knn", "gate", "node_gemm", "node_query", "edge_x2h", "edge_h2x", "edge_x2h_listed",
                   "edge_x2h_bwd", "edge_h2x_bwd", "train_gemm", "edge_x2h_bwd_listed", "edge_rows_reduce")


class NativeError(RuntimeError):
    pass


def _load(path, extra=None):
    dll = ctypes.CDLL(path)
    exports = dict(EXPORTS)
    exports.update(extra or {})
    for name, (res, args) in exports.items():
        fn = getattr(dll, name)  # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    if dll.cbgx_abi_version() != ABI_VERSION:
        raise NativeError(f"{path}: ABI version {dll.cbgx_abi_version()} != {ABI_VERSION} (stale build? run `python -m cbgbench_amd.build`)")
    return dll


_OVERRIDE = contextvars.ContextVar("cbgx_library_override", default=None)


def lib():
    """Load libcbgx.so (once). Raises NativeError if it has not been built."""
    global _LIB
    over = _OVERRIDE.get()
    if over is not None:      # inside `first_generation_kernels()` (tests only): this context's calls go to libcbgx_xcheck.so
        return over
    if _LIB is None:
        # CBGX_LIBRARY: another build of the same library (scripts/abl_bwd.sh points it at libcbgx_ablate.so)
        path = os.environ.get("CBGX_LIBRARY", LIBPATH)
        if not os.path.exists(path):
            raise NativeError(
                f"{path} not found: build it with `python -m cbgbench_amd.build` "
                "(there is no CPU / PyTorch fallback for the message-passing path)")
        _LIB = _load(path)
    return _LIB


@contextlib.contextmanager
def first_generation_kernels(impl=1):
    """TEST-ONLY: inside the block every binding call of THIS context (contextvars: thread / task local, no module global is
    swapped) goes to libcbgx_xcheck.so (include/cbgx_xcheck.h) with the first-generation VALU kernels selected (impl=2: the
    current kernels except the x2h backward, which is the second-generation one) -- an independent on-device implementation
    of the same stages.  The product library has neither those kernels nor the switch.  The kernel selection inside
    libcbgx_xcheck.so is process-wide, so concurrent blocks with different ``impl`` are not supported (tests run them serially)."""
    global _XLIB
    if _XLIB is None:
        if not os.path.exists(XCHECK_LIBPATH):
            raise NativeError(f"{XCHECK_LIBPATH} not found: build it with `python -m cbgbench_amd.build`")
        _XLIB = _load(XCHECK_LIBPATH, {"cbgx_debug_set_edge_kernel": (_i, [_i])})
    old = _XLIB.cbgx_debug_set_edge_kernel(impl)
    token = _OVERRIDE.set(_XLIB)
    try:
        yield _XLIB
    finally:
        _OVERRIDE.reset(token)
        _XLIB.cbgx_debug_set_edge_kernel(old)


def check(rc, what):
    if rc != 0:
        msg = lib().cbgx_last_error().decode(errors="replace")
        if rc == -1:
            raise ValueError(f"{what}: {msg}")
        raise NativeError(f"{what} failed ({rc}): {msg}")


def ptr(t):
    """Device pointer of a contiguous tensor (None -> NULL)."""
    if t is None:
        return None
    if not t.is_contiguous():
        raise ValueError("libcbgx needs contiguous tensors")
    return ctypes.c_void_p(t.data_ptr())


def current_stream(device):
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)



def version(t):
    """the in-place-write counter of a tensor, for caches keyed on "same tensor, not written since".  Inference tensors
    (created under torch.inference_mode()) keep no counter and reading it raises; they get a value that never compares equal, even to
    itself across two calls, so every cache treats them as always stale and takes its uncached route."""
    if t.is_inference():
        return _NeverEqual()
    return t._version


class _NeverEqual:
    __slots__ = ()

    def __eq__(self, other):
        return False

    def __ne__(self, other):
        return True

    def __hash__(self):
        return id(self)
