"""ctypes loader for libfsf_hip.so — the only way the package reaches the GPU kernels.

There is deliberately NO CPU fallback: if the library is missing or a call is made without a HIP device the
product path raises.  (The CPU oracle under `oracle/` is test infrastructure and is never imported here.)
"""
import ctypes
import os
import threading

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# FSF_LIB_PATH: another BUILD of the same library (same ABI version, checked below) — same-box A/B of two source states by
# tools/profiling/ab_bench.sh; never set in product runs
LIB_PATH = os.environ.get("FSF_LIB_PATH") or os.path.join(_HERE, "libfsf_hip.so")

_lib = None
_lock = threading.Lock()

c_i32 = ctypes.c_int32
c_i64 = ctypes.c_int64
c_f32 = ctypes.c_float
c_p = ctypes.c_void_p


class FsfHipError(RuntimeError):
    pass


def _header_abi_version():
    """FSF_ABI_VERSION of include/fsf_hip.h (None when the header does not travel with the package)."""
    import re

    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "fsf_hip.h")
    try:
        with open(path) as f:
            m = re.search(r"#define\s+FSF_ABI_VERSION\s+(\d+)", f.read())
        return int(m.group(1)) if m else None
    except OSError:
        return None


def lib():
    """Load (once) and return the ctypes handle.  Fails loudly when the extension has not been built."""
    global _lib
    if _lib is None:
        with _lock:
            if _lib is None:
                if not os.path.exists(LIB_PATH):
                    raise FsfHipError(
                        f"{LIB_PATH} is missing: the HIP extension was not built. "
                        "Run `python -m fullysparsefusion_amd.build` (needs hipcc); there is no CPU fallback."
                    )
                h = ctypes.CDLL(LIB_PATH)
                want = _header_abi_version()
                if want is not None and int(h.fsf_abi_version()) != want:
                    raise FsfHipError(f"{LIB_PATH} has ABI version {int(h.fsf_abi_version())}, include/fsf_hip.h declares {want}: "
                                      "stale build, run `python -m fullysparsefusion_amd.build`")
                h.fsf_status_string.restype = ctypes.c_char_p
                h.fsf_status_string.argtypes = [ctypes.c_int]
                for name in (
                    "fsf_unique_rows_workspace_bytes", "fsf_sir_stack_arena_bytes",
                    "fsf_segment_plan_workspace_bytes",
                    "fsf_segment_reduce_workspace_bytes",
                    "fsf_rulebook_workspace_bytes", "fsf_rulebook_to_pairs_workspace_bytes",
                    "fsf_ingroup_rank_workspace_bytes", "fsf_dynamic_point_pool_workspace_bytes",
                    "fsf_nms_bev_workspace_bytes", "fsf_nms_bev_multiclass_workspace_bytes", "fsf_nms_bev_multiclass_capped_workspace_bytes",
                    "fsf_norm_act_backward_workspace_bytes", "fsf_column_stats_workspace_bytes",
                    "fsf_connected_components_workspace_bytes",
                    "fsf_spconv_workspace_bytes", "fsf_spconv_backward_weight_workspace_bytes",
                    "fsf_linear_prepared_weight_bytes", "fsf_linear_prepared_weight_sliced_bytes", "fsf_spconv_split_weight_bytes", "fsf_spconv_split_workspace_bytes",
                    "fsf_planes_bytes", "fsf_planes_scale_count", "fsf_spconv_planes_weight_bytes", "fsf_assemble_sweeps_workspace_bytes",
                    "fsf_get_option", "fsf_order_by_neighbor_mask_workspace_bytes",
                    "fsf_class_rank_desc_workspace_bytes", "fsf_nms_select_capacity", "fsf_cluster_key_survival_workspace_bytes",
                    "fsf_overlap_plan_workspace_bytes", "fsf_group_pairs_workspace_bytes",
                    "fsf_row_planes_bytes", "fsf_linear_prepared_weight_f16_bytes", "fsf_spconv_split_weight_f16_bytes",
                ):
                    getattr(h, name).restype = c_i64
                _lib = h
    return _lib


def check(status, what):
    if status != 0:
        msg = lib().fsf_status_string(int(status)).decode()
        err = FsfHipError(f"{what} failed: {msg} (status {status})")
        err.status = int(status)
        raise err


def require_cuda(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise FsfHipError(
                "fullysparsefusion_amd ops run on the HIP device only (got a CPU tensor); there is no CPU fallback"
            )


def ptr(t):
    """Device pointer of a contiguous tensor (None -> NULL)."""
    if t is None:
        return c_p(None)
    if not t.is_contiguous():
        raise FsfHipError("non-contiguous tensor passed to the C ABI")
    return c_p(t.data_ptr())


def _raw_stream(device_index=None):
    """hipStream_t of torch's current stream as an int.  `torch.cuda.current_stream()` builds a Stream object (~8 us, and
    the wrappers need it ~300 times per frame); the raw getter is the same value without the object."""
    if device_index is None:
        device_index = torch.cuda.current_device()
    return torch._C._cuda_getCurrentRawStream(device_index)


def stream_ptr():
    return c_p(_raw_stream())


_ws_cache = {}


def workspace(nbytes, device):
    """Grow-only scratch buffer per (device, stream).  Stream-ordered reuse is safe because every C-ABI
    call enqueues all of its work on the current stream before returning."""
    index = device.index if device.index is not None else torch.cuda.current_device()
    key = (index, _raw_stream(index))
    buf = _ws_cache.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(int(nbytes), 1 << 20), dtype=torch.uint8, device=device)
        _ws_cache[key] = buf
    return buf


def f32_array(vals):
    return (c_f32 * len(vals))(*[float(v) for v in vals])


def i32_array(vals):
    return (c_i32 * len(vals))(*[int(v) for v in vals])


def i64_array(vals):
    return (c_i64 * len(vals))(*[int(v) for v in vals])
