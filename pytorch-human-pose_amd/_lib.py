"""ctypes binding of csrc/libhhrnet.so (the C-ABI declared in include/hhrnet.h).

There is no CPU fallback: if the HIP library cannot be loaded every product entry point
raises.  `build()` compiles it in-tree with hipcc for gfx950 (works without a GPU).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
SO = os.environ.get("HH_LIB") or os.path.join(CSRC, "libhhrnet.so")  # HH_LIB: A/B-test another build of the same ABI
HEADER = os.path.join(os.path.dirname(_HERE), "include", "hhrnet.h")
_lib = None


class HHError(RuntimeError):
    pass


def build(force: bool = False, jobs: int = 8) -> str:
    args = ["make", "-C", CSRC, f"-j{jobs}"]
    if force:
        args.append("-B")
    subprocess.check_call(args, stdout=subprocess.DEVNULL)
    return SO


def _sig(lib):
    i32, i64, dbl, vp, cp = C.c_int, C.c_int64, C.c_double, C.c_void_p, C.c_char_p
    pi64 = C.POINTER(C.c_int64)
    sigs = {
        "hh_abi_version": (i32, []),
        "hh_last_error": (cp, []),
        "hh_create": (vp, [i32, i32, i32]),
        "hh_destroy": (None, [vp]),
        "hh_create_classifier": (vp, [i32, i32, i32]),
        "hh_forward_classifier": (i32, [vp, vp, i32, i32, i32, vp, vp]),
        "hh_num_params": (i32, [vp]),
        "hh_param_name": (cp, [vp, i32]),
        "hh_param_shape": (i32, [vp, i32, pi64]),
        "hh_load_weights": (i32, [vp, cp, vp, pi64, i32]),
        "hh_finalize": (i32, [vp]),
        "hh_calibrate": (i32, [vp, vp, i32, i32, i32, i32, vp]),
        "hh_e4m3_encode": (i32, [vp, i64, vp]),
        "hh_e4m3_decode": (i32, [vp, i64, vp]),
        "hh_reserve": (i32, [vp, i32, i32, i32]),
        "hh_workspace_bytes": (i64, [vp]),
        "hh_forward": (i32, [vp, vp, i32, i32, i32, vp, vp, i32, vp]),
        "hh_forward_flops": (dbl, [vp, i32, i32, i32]),
        "hh_set_taps": (i32, [vp, i32]),
        "hh_set_multi_lane": (i32, [vp, i32]),
        "hh_num_taps": (i32, [vp]),
        "hh_tap_name": (cp, [vp, i32]),
        "hh_tap_shape": (i32, [vp, i32, pi64]),
        "hh_tap_read": (i32, [vp, i32, vp]),
        "hh_profile_enable": (i32, [vp, i32]),
        "hh_profile_count": (i32, [vp]),
        "hh_profile_get": (i32, [vp, i32, C.POINTER(C.c_int), C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_float), C.POINTER(C.c_float),
                           C.POINTER(C.c_char_p)]),
        "hh_profile_clock": (i32, [vp, i32, C.POINTER(C.c_double)]),
        "hh_conv_config": (i32, [i32, C.POINTER(C.c_int)]),
        "hh_conv_config_double_buffered": (i32, [i32]),
        "hh_debug_munkres": (i32, [C.POINTER(C.c_double), i32, C.POINTER(C.c_int32)]),
        "hh_debug_conv_bench": (i32, [i32, i32, i32, i32, i32, i32, i32, i32, i32, C.POINTER(C.c_float), vp, i32, C.POINTER(C.c_float)]),
        "hh_debug_bb_bench": (i32, [i32, i32, i32, i32, C.POINTER(C.c_float), vp]),
        "hh_debug_bb_compare": (i32, [i32, i32, i32, i32, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_float)]),
        "hh_preprocess_u8": (i32, [vp, i32, i32, C.POINTER(C.c_double), vp, i32, i32, C.POINTER(C.c_float), C.POINTER(C.c_float), vp]),
        "hh_preprocess_u8_batch": (i32, [vp, vp, i32, vp, i32, i32, C.POINTER(C.c_float), C.POINTER(C.c_float), vp]),
        "hh_flip_images": (i32, [vp, vp, i32, i32, i32, i32, vp]),
        "hh_flip_merge": (i32, [vp, i64, vp, i64, vp, i64, vp, i64, vp, i32, i32, i32, i32, vp]),
        "hh_decoder_create": (vp, [i32, i32, dbl, dbl]),
        "hh_decoder_destroy": (None, [vp]),
        "hh_decoder_reserve": (i32, [vp, i32, i32, i32, i32]),
        "hh_decoder_set_exact_topk": (i32, [vp, i32]),
        "hh_decode": (i32, [vp, vp, i64, vp, i64, vp, pi64, i32, i32, i32, i32, i32, i32, vp, vp, vp, vp, vp]),
        "hh_parse": (i32, [vp, vp, vp, i32, i32, i32, i32, i32, i32, vp, vp, vp, vp, vp]),
        "hh_debug_check_plan": (i32, [vp]),
        "hh_loss_heatmaps": (i32, [vp, i64, vp, vp, i32, i32, i32, i32, vp, vp, i64, vp, vp]),
        "hh_loss_ae_grouping": (i32, [vp, i64, vp, vp, i32, i32, i32, i32, i32, vp, vp, i64, C.c_float, C.c_float, vp, vp]),
        "hh_conv2d_workspace_bytes": (i64, [i32, i32, i32, i32]),
        "hh_conv2d": (i32, [vp, i32, i32, i32, i32, vp, i32, i32, i32, i32, i32, i32, vp, vp, i32, vp, vp, vp]),
        "hh_conv2d_wgrad_workspace_bytes": (i64, [i32, i32, i32, i32, i32, i32, i32]),
        "hh_conv2d_wgrad": (i32, [vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, vp, vp, vp]),
        "hh_fusion_sum_forward": (i32, [vp, vp, i32, i32, i32, i32, i32, i32, vp, vp]),
        "hh_fusion_sum_backward": (i32, [vp, vp, i32, i32, i32, i32, i32, vp, vp, vp, i32, vp]),
        "hh_bn_train_forward": (i32, [vp, i64, i32, vp, vp, C.c_float, vp, i32, vp, vp, vp, vp, vp]),
        "hh_bn_train_backward": (i32, [vp, vp, vp, i64, i32, vp, vp, vp, i32, vp, vp, vp, vp, vp, vp]),
        "hh_bn_train_backward_plain": (i32, [vp, vp, i64, i32, vp, vp, vp, vp, i32, vp, vp, vp, vp, vp]),
        "hh_conv2d_packed_elems": (i64, [i32, i32, i32, i32, i32]),
        "hh_pack_conv_weights_batch": (i32, [i32, vp, vp, vp, vp, vp]),
        "hh_conv2d_packed": (i32, [vp, i32, i32, i32, i32, vp, i32, i32, i32, i32, i32, i32, vp, vp, i32, vp, vp]),
        "hh_bn_train_stats": (i32, [vp, i64, i32, vp, vp, vp]),
        "hh_bn_train_normalize": (i32, [vp, i64, i32, vp, dbl, vp, vp, C.c_float, vp, i32, vp, vp, vp, vp]),
        "hh_bn_train_backward_stats": (i32, [vp, vp, vp, i64, i32, vp, vp, i32, vp, vp, vp, vp, vp]),
        "hh_bn_train_backward_apply": (i32, [vp, vp, vp, i64, i32, vp, vp, vp, i32, vp, dbl, vp, vp, vp, vp]),
        "hh_resize_accumulate": (i32, [vp, i64, i32, i32, i32, i32, vp, i64, i32, i32, C.c_float, i32, vp]),
        "hh_decoder_read_topk": (i32, [vp, vp, vp, vp]),
        "hh_transform_coords": (i32, [vp, i32, dbl, dbl, dbl, dbl, dbl, vp]),
        "hh_get_affine_transform": (i32, [dbl, dbl, dbl, dbl, dbl, i32, C.POINTER(C.c_double)]),
        "hh_invert_affine": (i32, [C.POINTER(C.c_double), C.POINTER(C.c_double)]),
        "hh_warp_affine_u8": (i32, [vp, i32, i32, C.POINTER(C.c_double), vp, i32, i32, vp]),
    }
    for name, (res, args) in sigs.items():
        if name.startswith("hh_debug_") and not hasattr(lib, name):
            continue  # (test / probe hooks only: an older build of the same ABI behind HH_LIB may lack one)
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    return sigs


def exported_symbols() -> list[str]:
    """Entry points declared in include/hhrnet.h (parsed from the header text)."""
    import re
    txt = open(HEADER).read()
    return sorted(set(re.findall(r"\b(hh_[a-z_0-9]+)\s*\(", txt)))


def load() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(SO):
            # Normally __graft_entry__.build() has produced the library.  As a convenience a missing one is built here, under an
            # exclusive file lock so that the ranks of a torchrun launch do not compile into the same tree at once (the first
            # holder builds, the others find the finished file), and never once this process has touched the GPU (a compiler
            # child process next to a live HIP context is what the GPU boxes forbid under rocprofv3).
            import fcntl
            import torch
            if torch.cuda.is_initialized():
                raise HHError(f"{SO} is missing; run __graft_entry__.build() before any GPU work")
            with open(os.path.join(CSRC, ".build.lock"), "w") as lock:
                fcntl.flock(lock, fcntl.LOCK_EX)
                try:
                    if not os.path.exists(SO):
                        build()
                except Exception as e:  # noqa: BLE001
                    raise HHError(f"{SO} is missing and could not be built ({e}); run __graft_entry__.build()") from e
                finally:
                    fcntl.flock(lock, fcntl.LOCK_UN)
        lib = C.CDLL(SO)
        _sig(lib)
        if lib.hh_abi_version() != 3:
            raise HHError("libhhrnet.so ABI version mismatch")
        _lib = lib
    return _lib


def check(rc: int) -> None:
    if rc != 0:
        raise HHError(load().hh_last_error().decode())
