"""ctypes binding of csrc/libgsrast_mi355x.so (C ABI: include/gsrast_mi355x.h).

The HIP library is the product path.  There is deliberately NO fallback: if the
shared object is missing, cannot be loaded, or a tensor is not on a HIP device,
the operators raise.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# GSR_LIBRARY selects another build of the same library (tools/ab_build.sh + tools/ab_run.sh: A/B variants live in their own directory, the
# shipped .so is never rebuilt in place); gsr_build_info() says what a loaded binary was built with.
SO_PATH = os.environ.get("GSR_LIBRARY") or os.path.join(_HERE, "csrc", "libgsrast_mi355x.so")
ABI_VERSION = 16

_lib = None

c_void_p = ctypes.c_void_p
c_int = ctypes.c_int
c_i64 = ctypes.c_int64
c_float = ctypes.c_float
c_size = ctypes.c_size_t

# name -> (restype, argtypes); mirrors include/gsrast_mi355x.h one to one
PROTOTYPES = {
    "gsr_abi_version": (c_int, []),
    "gsr_last_error": (ctypes.c_char_p, []),
    "gsr_geometry_bytes": (c_size, [c_int]),
    "gsr_image_bytes": (c_size, [c_int, c_int]),
    "gsr_binning_bytes": (c_size, [c_i64, c_int, c_int, c_int]),
    "gsr_backward_scratch_bytes": (c_size, [c_i64, c_int, c_int]),
    "gsr_gradient_row_limit": (c_i64, []),
    "gsr_gradient_rows": (c_int, [c_int, c_void_p, c_size, ctypes.POINTER(c_i64), c_void_p]),
    "gsr_forward_geometry": (c_int, [c_int, c_int, c_int, c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_void_p, c_void_p,
                                     c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_float, c_float, c_int, c_void_p,
                                     c_void_p, c_size, ctypes.POINTER(c_i64), c_void_p]),
    "gsr_forward_render": (c_int, [c_int, c_i64, c_int, c_int, c_void_p, c_void_p, c_size, c_void_p, c_size, c_void_p, c_size,
                                   c_void_p, c_void_p]),
    "gsr_forward": (c_int, [c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                            c_void_p, c_void_p, c_void_p, c_int, c_int, c_float, c_float, c_int, c_void_p, c_void_p, c_size, c_void_p,
                            c_size, c_void_p, c_size, c_void_p, ctypes.POINTER(c_i64), c_void_p]),
    "gsr_backward": (c_int, [c_int, c_int, c_int, c_i64, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_float,
                             c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_float, c_void_p, c_void_p, c_size,
                             c_void_p, c_size, c_void_p, c_size, c_void_p, c_size, c_void_p, c_void_p, c_void_p, c_void_p,
                             c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "gsr_forward_params": (c_int, [c_int, c_int, c_int] + [c_void_p] * 7 + [c_float, c_void_p, c_void_p, c_void_p, c_int, c_int,
                                   c_float, c_float, c_void_p, c_void_p, c_size, c_void_p, c_size, c_void_p, c_size, c_void_p,
                                   ctypes.POINTER(c_i64), c_void_p]),
    "gsr_backward_params": (c_int, [c_int, c_int, c_int, c_i64, c_int, c_int] + [c_void_p] * 6 + [c_float, c_void_p, c_void_p,
                                    c_void_p, c_float, c_float, c_void_p, c_void_p, c_size, c_void_p, c_size, c_void_p, c_size,
                                    c_void_p, c_size] + [c_void_p] * 9),
    "gsr_adam_step": (c_int, [c_int, c_void_p, ctypes.c_double, ctypes.c_double, ctypes.c_double, c_void_p]),
    "gsr_densification_stats": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "gsr_compact_scratch_bytes": (c_size, [c_int]),
    "gsr_compact_plan": (c_int, [c_int, c_void_p, c_void_p, c_size, ctypes.POINTER(c_i64), c_void_p]),
    "gsr_compact_rows": (c_int, [c_int, c_void_p, c_void_p, c_size, c_int, c_void_p, c_void_p]),
    "gsr_mark_visible": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "gsr_loss_scratch_bytes": (c_size, [c_int, c_int, c_int]),
    "gsr_loss_forward": (c_int, [c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_size, c_void_p, c_void_p]),
    "gsr_loss_backward": (c_int, [c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_size, c_void_p, c_void_p,
                                  c_void_p, c_void_p]),
    "gsr_appearance_saved_bytes": (c_size, [c_int, c_int]),
    "gsr_appearance_scratch_bytes": (c_size, [c_int, c_int]),
    "gsr_appearance_tail_forward": (c_int, [c_int, c_int, c_int, c_int] + [c_void_p] * 8 + [c_void_p, c_size, c_void_p]),
    "gsr_appearance_tail_backward": (c_int, [c_int, c_int, c_int, c_int] + [c_void_p] * 5 + [c_void_p, c_size] + [c_void_p] * 8 +
                                     [c_void_p, c_size, c_int, c_void_p]),
    "gsr_appearance_up_scratch_bytes": (c_size, [c_int, c_int]),
    "gsr_appearance_up_forward": (c_int, [c_int, c_int] + [c_void_p] * 4 + [c_void_p]),
    "gsr_appearance_up_backward": (c_int, [c_int, c_int] + [c_void_p] * 6 + [c_void_p, c_size, c_void_p]),
    "gsr_bias_relu_scratch_bytes": (c_size, [c_int]),
    "gsr_bias_relu_forward": (c_int, [c_int, c_i64, c_void_p, c_void_p, c_void_p]),
    "gsr_bias_relu_backward": (c_int, [c_int, c_i64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size, c_void_p]),
    "gsr_mse_scratch_bytes": (c_size, [c_int]),
    "gsr_mse": (c_int, [c_int, ctypes.c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_size, c_void_p]),
    "gsr_knn_scratch_bytes": (c_size, [c_int]),
    "gsr_knn_mean_dist2": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_size, c_void_p]),
    "gsr_debug_export": (c_int, [c_int, c_i64, c_int, c_int] + [c_void_p] * 17),
    "gsr_debug_capture_sums": (c_int, [c_void_p]),
    "gsr_debug_knn_counter": (c_int, [c_void_p]),
    "gsr_debug_blend_profile": (c_int, [ctypes.POINTER(ctypes.c_uint64)]),
    "gsr_build_info": (ctypes.c_char_p, []),
    "gsr_set_variant": (c_int, [ctypes.c_char_p, c_int]),
    "gsr_get_variant": (c_int, [ctypes.c_char_p, c_int]),
    "gsr_profile_enable": (None, [c_int]),
    "gsr_profile_select": (None, [ctypes.c_uint32]),
    "gsr_profile_collect": (c_int, [ctypes.POINTER(ctypes.c_double), ctypes.POINTER(c_i64), c_int]),
    "gsr_sort_scratch_bytes": (c_size, [c_i64]),
    "gsr_sort_pairs": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_i64, c_int, c_void_p, c_size, c_void_p]),
}


class AdamTensor(ctypes.Structure):
    """gsr_adam_tensor of include/gsrast_mi355x.h"""
    _fields_ = [("param", c_void_p), ("grad", c_void_p), ("exp_avg", c_void_p), ("exp_avg_sq", c_void_p), ("n", c_i64),
                ("lr", ctypes.c_double), ("step", c_i64)]


class CompactTensor(ctypes.Structure):
    """gsr_compact_tensor of include/gsrast_mi355x.h"""
    _fields_ = [("src", c_void_p), ("dst", c_void_p), ("row_floats", c_i64)]


class GsrError(RuntimeError):
    """Raised for every non-zero status of the C ABI (the reference throws std::runtime_error / AT_ERROR)."""


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(SO_PATH):
            raise GsrError(f"HIP extension not built: {SO_PATH} is missing (run `python -c 'import __graft_entry__ as g; g.build()'` "
                           "or `make -C based_on_point_cloud_partitions_amd/csrc`). There is no CPU fallback.")
        lib = ctypes.CDLL(SO_PATH)
        for name, (res, args) in PROTOTYPES.items():
            fn = getattr(lib, name)  # AttributeError here = the .so does not export what the header declares
            fn.restype = res
            fn.argtypes = args
        if lib.gsr_abi_version() != ABI_VERSION:
            raise GsrError(f"ABI mismatch: library {lib.gsr_abi_version()} != binding {ABI_VERSION}")
        _lib = lib
    return _lib


STAGES = ["preprocess", "scan", "emit_keys", "radix_sort", "tile_ranges", "render_forward", "render_backward",
          "gaussian_backward"]


def profile_enable(on, stages=None):
    """Per-stage HIP-event timing on/off; `stages` = iterable of stage names to restrict it to (None = all)."""
    lib = load()
    mask = 0xFFFFFFFF if stages is None else sum(1 << STAGES.index(n) for n in stages)
    lib.gsr_profile_select(mask)
    lib.gsr_profile_enable(1 if on else 0)


def profile_collect():
    """-> {stage: (total_ms, launches)} since the last collect."""
    n = len(STAGES)
    ms = (ctypes.c_double * n)()
    cnt = (c_i64 * n)()
    check(load().gsr_profile_collect(ms, cnt, n), "gsr_profile_collect")
    return {STAGES[i]: (float(ms[i]), int(cnt[i])) for i in range(n)}


def build_info():
    """gsr_build_info(): ABI version, the two compile-time ablation macros (0 in a shipped build) and make's EXTRA flags."""
    return load().gsr_build_info().decode()


def capture_sums(tensor):
    """Test hook (two-stage gradient check): while set, every backward also stores the per-Gaussian blend sums its chain consumes
    into `tensor` ([P,12] float32 on the device); None switches it off."""
    check(load().gsr_debug_capture_sums(None if tensor is None else tensor.data_ptr()), "gsr_debug_capture_sums")


def knn_counter(tensor):
    """Measurement hook: while set, distCUDA2 adds the number of candidate distances it evaluates to `tensor` (one int64 on the
    device, zeroed by the caller); None switches it off."""
    check(load().gsr_debug_knn_counter(None if tensor is None else tensor.data_ptr()), "gsr_debug_knn_counter")


def set_variant(name, value):
    """Overrides the experiment switch GSR_<name> for this process (tests / A-B runs)."""
    check(load().gsr_set_variant(name.encode(), int(value)), "gsr_set_variant")


def get_variant(name, default):
    """Current value of the experiment switch GSR_<name> (`default` if it was never set)."""
    return int(load().gsr_get_variant(name.encode(), int(default)))


def check(status, what):
    if status < 0:
        msg = load().gsr_last_error()
        raise GsrError(f"{what} failed ({status}): {msg.decode() if msg else ''}")
    return status


# ---- host-side fast paths (a rasterizer step is two C-ABI calls; on a 10k-Gaussian scene the Python around them is as long as the
# kernels) -- GSR_FAST_HOST=0 switches back to the plain torch calls
_FAST_HOST = os.environ.get("GSR_FAST_HOST", "1") != "0"


class _NoGuard:
    def __enter__(self):
        return None

    def __exit__(self, *exc):
        return False


_NO_GUARD = _NoGuard()


def device_guard(device):
    """`with torch.cuda.device(device)` -- but nothing at all when `device` already is the current one (the usual case: one process
    per GPU), which saves the two device switches of the context manager."""
    import torch

    if _FAST_HOST and device.index is not None and hasattr(torch._C, "_cuda_getDevice") and torch._C._cuda_getDevice() == device.index:
        return _NO_GUARD
    return torch.cuda.device(device)


def stream_ptr(device):
    """The raw handle of PyTorch's current stream on `device` as a ctypes pointer (without building a torch.cuda.Stream object)."""
    import torch

    if _FAST_HOST and device.index is not None and hasattr(torch._C, "_cuda_getCurrentRawStream"):  # (2.6 us -> 0.4 us for the two)
        return ctypes.c_void_p(torch._C._cuda_getCurrentRawStream(device.index))
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)
