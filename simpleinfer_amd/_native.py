"""ctypes loader for the two in-tree native libraries.

There is no Python or CPU fallback: if the libraries are missing (not built) this raises, and on a
machine without a HIP device every compute entry point returns an error code that the wrappers turn
into an exception.
"""
from __future__ import annotations

import ctypes as C
import os

PKG = os.path.dirname(os.path.abspath(__file__))
# SI_HIP_LIB: a differently built kernel library (development: diagnostic / variant builds under build_variants/)
LIB_HIP_PATH = os.environ.get("SI_HIP_LIB") or os.path.join(PKG, "libsi_hip.so")
# SI_HOST_LIB: a differently built host library (tools/asan_host.sh points it at the ASan + UBSan build)
LIB_HOST_PATH = os.environ.get("SI_HOST_LIB") or os.path.join(PKG, "libsimpleinfer_amd.so")

_hip = None
_host = None


class NativeLibraryMissing(ImportError):
    pass


def _load(path):
    if not os.path.exists(path):
        raise NativeLibraryMissing(
            "%s is not built: run `python -m simpleinfer_amd.build` (or __graft_entry__.build()); "
            "simpleinfer_amd has no non-HIP fallback" % path)
    return C.CDLL(path, mode=C.RTLD_GLOBAL)


class SiConvPlan(C.Structure):
    """include/si_hip.h SiConvPlan: kernel-form choices of one call (every form of a family produces the same bits)"""
    _fields_ = [(k, C.c_int) for k in ("f32_tile", "wino23_form", "wino23_ocg", "f16_tile", "f16_detect_tile", "f16_s2c32", "f16_slab",
                                       "f16_slab_w2", "f16_pw_patch", "split3_bm")]
    DEFAULTS = dict(f32_tile=-1, wino23_form=0, wino23_ocg=0, f16_tile=-1, f16_detect_tile=-1, f16_s2c32=-1, f16_slab=-1, f16_slab_w2=-1,
                    f16_pw_patch=-1, split3_bm=0)

    def __init__(self, **kw):
        super().__init__()
        vals = dict(self.DEFAULTS)
        for k, v in kw.items():
            if k not in vals:
                raise TypeError("SiConvPlan has no field %r" % k)
            vals[k] = int(v)
        for k, v in vals.items():
            setattr(self, k, v)


class SiConv2dDesc(C.Structure):
    # (fewer positional initialisers than fields leave the trailing ones -- plan, range_flag -- NULL: the defaults)
    _fields_ = [(k, C.c_int) for k in
                ("n", "ih", "iw", "ic", "in_ld", "oh", "ow", "oc", "out_ld", "kh", "kw", "sh", "sw", "dh", "dw",
                 "pt", "pl", "groups", "has_bias", "act1", "has_residual", "res_ld", "act2")] + [
                     ("act_param", C.c_float), ("plan", C.POINTER(SiConvPlan)), ("range_flag", C.c_void_p)]


class SiConv2dUpsampledSource(C.Structure):
    _fields_ = [("src", C.c_void_p), ("ih", C.c_int), ("iw", C.c_int), ("c", C.c_int), ("ld", C.c_int), ("c0", C.c_int),
                ("inv_scale_h", C.c_float), ("inv_scale_w", C.c_float)]


class SiYoloLevel(C.Structure):
    _fields_ = [("na", C.c_int), ("ne", C.c_int), ("rows_total", C.c_int), ("row_off", C.c_int), ("stride", C.c_float)]


class SiPool2dDesc(C.Structure):
    _fields_ = [(k, C.c_int) for k in
                ("n", "ih", "iw", "c", "in_ld", "oh", "ow", "out_ld", "kh", "kw", "sh", "sw", "dh", "dw", "pt", "pl")]


def hip():
    """libsi_hip.so with argtypes/restypes set (include/si_hip.h)."""
    global _hip
    if _hip is not None:
        return _hip
    L = _load(LIB_HIP_PATH)
    vp, sz, i, f = C.c_void_p, C.c_size_t, C.c_int, C.c_float
    ip = C.POINTER(C.c_int)
    sig = {
        "si_hip_version": (C.c_char_p, []),
        "si_hip_error_string": (C.c_char_p, [i]),
        "si_hip_device_count": (i, [ip]),
        "si_hip_set_device": (i, [i]),
        "si_hip_get_device": (i, [ip]),
        "si_hip_device_info": (i, [i, C.c_char_p, ip, C.POINTER(sz), ip]),
        "si_hip_malloc": (i, [C.POINTER(vp), sz]),
        "si_hip_free": (i, [vp]),
        "si_hip_host_alloc": (i, [C.POINTER(vp), sz]),
        "si_hip_host_free": (i, [vp]),
        "si_hip_host_register": (i, [vp, sz]),
        "si_hip_host_unregister": (i, [vp]),
        "si_hip_memset_async": (i, [vp, i, sz, vp]),
        "si_hip_memcpy_h2d": (i, [vp, vp, sz, vp]),
        "si_hip_memcpy_d2h": (i, [vp, vp, sz, vp]),
        "si_hip_memcpy_d2d": (i, [vp, vp, sz, vp]),
        "si_hip_stream_create": (i, [C.POINTER(vp)]),
        "si_hip_stream_create_priority": (i, [C.POINTER(vp), i]),
        "si_hip_stream_destroy": (i, [vp]),
        "si_hip_stream_sync": (i, [vp]),
        "si_hip_device_sync": (i, []),
        "si_hip_event_create": (i, [C.POINTER(vp)]),
        "si_hip_event_destroy": (i, [vp]),
        "si_hip_event_record": (i, [vp, vp]),
        "si_hip_event_sync": (i, [vp]),
        "si_hip_event_elapsed_ms": (i, [vp, vp, C.POINTER(f)]),
        "si_hip_stream_wait_event": (i, [vp, vp]),
        "si_hip_ipc_get_mem_handle": (i, [vp, vp]),
        "si_hip_ipc_open_mem_handle": (i, [vp, C.POINTER(vp)]),
        "si_hip_ipc_close_mem_handle": (i, [vp]),
        "si_hip_enable_peer_access": (i, [i]),
        "si_hip_device_pci_bus_id": (i, [i, C.c_char_p, i]),
        "si_hip_device_by_pci_bus_id": (i, [C.c_char_p]),
        "si_hip_graph_begin_capture": (i, [vp]),
        "si_hip_graph_end_capture": (i, [vp, C.POINTER(vp)]),
        "si_hip_graph_launch": (i, [vp, vp]),
        "si_hip_graph_destroy": (i, [vp]),
        "si_hip_conv2d_weight_elems": (sz, [C.POINTER(SiConv2dDesc)]),
        "si_hip_conv2d_pack_weight_host": (i, [C.POINTER(SiConv2dDesc), vp, vp]),
        "si_hip_conv2d_f32": (i, [C.POINTER(SiConv2dDesc), vp, vp, vp, vp, vp, vp]),
        "si_hip_conv2d_split3_supported": (i, [C.POINTER(SiConv2dDesc)]),
        "si_hip_conv2d_split3_weight_elems": (sz, [C.POINTER(SiConv2dDesc)]),
        "si_hip_conv2d_split3_pack_weight_host": (i, [C.POINTER(SiConv2dDesc), vp, vp]),
        "si_hip_conv2d_split3_f32": (i, [C.POINTER(SiConv2dDesc), vp, vp, vp, vp, vp, vp]),
        "si_hip_conv2d_split3_split_f32": (i, [C.POINTER(SiConv2dDesc), vp, vp, vp, vp, i, vp, i, vp]),
        "si_hip_conv2d_split3_upcat_supported": (i, [C.POINTER(SiConv2dDesc), C.POINTER(SiConv2dUpsampledSource)]),
        "si_hip_conv2d_split3_upcat_f32": (i, [C.POINTER(SiConv2dDesc), vp, C.POINTER(SiConv2dUpsampledSource), vp, vp, vp, i, vp, i, vp]),
        "si_hip_conv2d_wino23_split_supported": (i, [C.POINTER(SiConv2dDesc)]),
        "si_hip_conv2d_wino23_split_weight_elems": (sz, [C.POINTER(SiConv2dDesc)]),
        "si_hip_conv2d_wino23_split_pack_weight_host": (i, [C.POINTER(SiConv2dDesc), vp, vp]),
        "si_hip_conv2d_wino23_split_f32": (i, [C.POINTER(SiConv2dDesc), vp, vp, vp, vp, vp, vp]),
        "si_hip_conv2d_wino23_eligible": (i, [C.POINTER(SiConv2dDesc)]),
        "si_hip_conv2d_wino23_preferred": (i, [C.POINTER(SiConv2dDesc)]),
        "si_hip_conv2d_wino23_weight_elems": (sz, [C.POINTER(SiConv2dDesc)]),
        "si_hip_conv2d_wino23_pack_weight_host": (i, [C.POINTER(SiConv2dDesc), vp, vp]),
        "si_hip_conv2d_wino23_f32": (i, [C.POINTER(SiConv2dDesc), vp, vp, vp, vp, vp, vp]),
        "si_hip_conv2d_wino43_eligible": (i, [C.POINTER(SiConv2dDesc)]),
        "si_hip_conv2d_wino43_preferred": (i, [C.POINTER(SiConv2dDesc)]),
        "si_hip_conv2d_wino43_weight_elems": (sz, [C.POINTER(SiConv2dDesc)]),
        "si_hip_conv2d_wino43_pack_weight_host": (i, [C.POINTER(SiConv2dDesc), vp, vp]),
        "si_hip_conv2d_wino43_f32": (i, [C.POINTER(SiConv2dDesc), vp, vp, vp, vp, vp, vp]),
        "si_hip_conv2d_split_f32": (i, [C.POINTER(SiConv2dDesc), vp, vp, vp, vp, i, vp, i, vp]),
        "si_hip_conv2d_upcat_f32": (i, [C.POINTER(SiConv2dDesc), vp, C.POINTER(SiConv2dUpsampledSource), vp, vp, vp, i, vp, i, vp]),
        "si_hip_conv2d_yolo_f32": (i, [C.POINTER(SiConv2dDesc), vp, vp, vp, C.POINTER(SiYoloLevel), vp, vp, vp, vp]),
        "si_hip_conv2d_split3_yolo_f32": (i, [C.POINTER(SiConv2dDesc), vp, vp, vp, C.POINTER(SiYoloLevel), vp, vp, vp, vp]),
        "si_hip_conv2d_upcat_supported": (i, [C.POINTER(SiConv2dDesc), C.POINTER(SiConv2dUpsampledSource)]),
        "si_hip_conv2d_kernel_name": (C.c_char_p, [C.POINTER(SiConv2dDesc), vp]),
        "si_hip_conv2d_kernel_name_form": (C.c_char_p, [C.POINTER(SiConv2dDesc), vp, i]),
        "si_hip_linear_f32": (i, [vp, i, i, vp, vp, i, vp, vp]),
        "si_hip_maxpool2d_f32": (i, [C.POINTER(SiPool2dDesc), vp, vp, vp]),
        "si_hip_adaptive_avgpool2d_f32": (i, [vp, i, i, i, i, i, vp, i, i, i, vp]),
        "si_hip_upsample_nearest_f32": (i, [vp, i, i, i, i, i, f, f, vp, i, i, i, vp]),
        "si_hip_copy_channels_f32": (i, [vp, sz, i, i, vp, i, vp]),
        "si_hip_cat_axis_f32": (i, [vp, ip, vp, ip, i, i, vp]),
        "si_hip_nhwc_to_nchw_f32": (i, [vp, i, i, i, i, i, vp, vp]),
        "si_hip_activation_f32": (i, [i, f, vp, sz, i, i, vp, i, vp]),
        "si_hip_binary_f32": (i, [i, vp, ip, i, vp, ip, i, vp, ip, i, vp]),
        "si_hip_binary_scalar_f32": (i, [i, vp, sz, i, i, f, vp, i, vp]),
        "si_hip_unary_f32": (i, [i, vp, sz, i, i, vp, i, vp]),
        "si_hip_batchnorm2d_f32": (i, [vp, sz, i, i, vp, vp, vp, vp, f, vp, i, vp]),
        "si_hip_yolo_decode_f32": (i, [vp, i, i, i, i, i, vp, vp, f, vp, i, i, vp]),
        "si_hip_f32_to_f16_host": (i, [vp, vp, sz]),
        "si_hip_f16_to_f32_host": (i, [vp, vp, sz]),
        "si_hip_conv2d_f16_supported": (i, [C.POINTER(SiConv2dDesc)]),
        "si_hip_conv2d_upcat_f16": (i, [C.POINTER(SiConv2dDesc), vp, C.POINTER(SiConv2dUpsampledSource), vp, vp, vp, i, vp, i, vp]),
        "si_hip_conv2d_upcat_f16_supported": (i, [C.POINTER(SiConv2dDesc), C.POINTER(SiConv2dUpsampledSource)]),
        "si_hip_conv2d_f16_tile_variant": (i, [C.POINTER(SiConv2dDesc)]),
        "si_hip_conv2d_f16_kernel_name": (C.c_char_p, [C.POINTER(SiConv2dDesc), i]),
        "si_hip_conv2d_f16_weight_elems": (sz, [C.POINTER(SiConv2dDesc)]),
        "si_hip_conv2d_f16_pack_weight_host": (i, [C.POINTER(SiConv2dDesc), vp, vp]),
        "si_hip_conv2d_f16": (i, [C.POINTER(SiConv2dDesc), vp, vp, vp, vp, vp, i, vp]),
        "si_hip_conv2d_depthwise_f16": (i, [C.POINTER(SiConv2dDesc), vp, vp, vp, vp, vp, vp]),
        "si_hip_conv2d_stem_f16": (i, [C.POINTER(SiConv2dDesc), vp, vp, vp, vp, vp]),
        "si_hip_conv2d_stem_f16_weight_elems": (sz, [C.POINTER(SiConv2dDesc)]),
        "si_hip_conv2d_stem_split3_f32": (i, [C.POINTER(SiConv2dDesc), vp, vp, vp, vp, vp]),
        "si_hip_conv2d_stem_split3_weight_elems": (sz, [C.POINTER(SiConv2dDesc)]),
        "si_hip_conv2d_stem_split3_pack_weight_host": (i, [C.POINTER(SiConv2dDesc), vp, vp]),
        "si_hip_conv2d_stem_f16_pack_weight_host": (i, [C.POINTER(SiConv2dDesc), vp, vp]),
        "si_hip_conv2d_split_f16": (i, [C.POINTER(SiConv2dDesc), vp, vp, vp, vp, i, vp, i, vp]),
        "si_hip_conv2d_yolo_f16": (i, [C.POINTER(SiConv2dDesc), vp, vp, vp, C.POINTER(SiYoloLevel), vp, vp, vp, vp]),
        "si_hip_conv2d_pw_slab_f16_supported": (i, [C.POINTER(SiConv2dDesc), C.POINTER(SiConv2dDesc)]),
        "si_hip_conv2d_pw_slab_f16": (i, [C.POINTER(SiConv2dDesc), C.POINTER(SiConv2dDesc), vp, vp, vp, vp, vp, vp, vp, vp]),
        "si_hip_conv2d_pw_cv3_f16_supported": (i, [C.POINTER(SiConv2dDesc), C.POINTER(SiConv2dDesc), C.POINTER(SiConv2dDesc)]),
        "si_hip_conv2d_pw_cv3_f16": (i, [C.POINTER(SiConv2dDesc), C.POINTER(SiConv2dDesc), C.POINTER(SiConv2dDesc), vp, vp, vp, vp, vp, vp, vp, i, vp, vp,
                                         vp, vp]),
        "si_hip_conv2d_stem_s2c32_f16_supported": (i, [C.POINTER(SiConv2dDesc), C.POINTER(SiConv2dDesc)]),
        "si_hip_conv2d_stem_s2c32_f16": (i, [C.POINTER(SiConv2dDesc), C.POINTER(SiConv2dDesc), vp, vp, vp, vp, vp, vp, vp]),
        "si_hip_conv2d_stem_s2c32_pw_f16_supported": (i, [C.POINTER(SiConv2dDesc), C.POINTER(SiConv2dDesc), C.POINTER(SiConv2dDesc), i]),
        "si_hip_conv2d_stem_s2c32_pw_f16": (i, [C.POINTER(SiConv2dDesc), C.POINTER(SiConv2dDesc), C.POINTER(SiConv2dDesc), vp, vp, vp, vp, vp, vp,
                                                vp, vp, i, vp, i, vp]),
        "si_hip_conv2d_yolo_f16_tile": (i, [C.POINTER(SiConv2dDesc), C.POINTER(SiYoloLevel)]),
        "si_hip_activation_f16": (i, [i, f, vp, sz, i, i, vp, i, vp]),
        "si_hip_unary_f16": (i, [i, vp, sz, i, i, vp, i, vp]),
        "si_hip_binary_same_f16": (i, [i, vp, i, vp, i, vp, i, sz, i, vp]),
        "si_hip_binary_bcast_f16": (i, [i, vp, i, vp, i, vp, i, i, sz, i, vp]),
        "si_hip_maxpool2d_f16": (i, [C.POINTER(SiPool2dDesc), vp, vp, vp]),
        "si_hip_maxpool5_chain3_f32": (i, [vp, i, i, i, i, i, vp, i, vp, i, vp, i, vp]),
        "si_hip_maxpool5_chain3_f16": (i, [vp, i, i, i, i, i, vp, i, vp, i, vp, i, vp]),
        "si_hip_adaptive_avgpool2d_f16": (i, [vp, i, i, i, i, i, vp, i, i, i, vp]),
        "si_hip_convert_f32_f16": (i, [vp, sz, i, i, vp, i, vp]),
        "si_hip_convert_f16_f32": (i, [vp, sz, i, i, vp, i, vp]),
        "si_letterbox_geometry": (None, [i, i, i, i, ip, ip, C.POINTER(f), ip, ip]),
        "si_hip_letterbox_u8_f32": (i, [vp, i, i, vp, i, i, i, i, vp]),
        "si_hip_letterbox_batch_u8_f32": (i, [vp, i, sz, i, i, vp, i, i, i, i, vp]),
        "si_hip_resize_bilinear_u8c3": (i, [vp, i, sz, i, i, vp, sz, i, i, vp]),
        "si_hip_resize_letterbox_batch_u8_f32": (i, [vp, i, sz, i, i, vp, i, i, vp]),
        "si_hip_yolo_postprocess_workspace_bytes": (sz, [i, i, i]),
        "si_hip_yolo_postprocess_f32": (i, [vp, i, i, i, f, f, i, vp, vp, vp, i, vp, sz, vp]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)  # AttributeError here = header/library mismatch, which tests check
        fn.restype = res
        fn.argtypes = args
    L._si_signatures = sig
    _hip = L
    return L


class SiGatherStats(C.Structure):
    """include/si_shard.h"""
    _fields_ = [("wait_copies_ms_total", C.c_double), ("wait_barrier_ms_total", C.c_double), ("copy_ms_total", C.c_double),
                ("copy_ms_max", C.c_double), ("completes", C.c_longlong), ("copies", C.c_longlong), ("landed_ms_total", C.c_double)]


def host():
    """libsimpleinfer_amd.so with argtypes/restypes set (include/si_engine.h)."""
    global _host
    if _host is not None:
        return _host
    hip()  # dependency, loaded RTLD_GLOBAL first
    L = _load(LIB_HOST_PATH)
    vp, sz, i = C.c_void_p, C.c_size_t, C.c_int
    cp = C.c_char_p
    sig = {
        "si_engine_create": (i, [C.POINTER(vp)]),
        "si_engine_destroy": (i, [vp]),
        "si_engine_set_option": (i, [vp, cp, i]),
        "si_engine_load_model": (i, [vp, cp, cp]),
        "si_engine_release": (i, [vp]),
        "si_engine_num_inputs": (i, [vp]),
        "si_engine_num_outputs": (i, [vp]),
        "si_engine_input_name": (cp, [vp, i]),
        "si_engine_output_name": (cp, [vp, i]),
        "si_engine_operand_shape": (i, [vp, cp, C.POINTER(i), C.POINTER(i)]),
        "si_engine_input": (i, [vp, cp, vp, i]),
        "si_engine_bind_output": (i, [vp, cp, vp]),
        "si_engine_forward": (i, [vp]),
        "si_engine_forward_async": (i, [vp]),
        "si_engine_sync": (i, [vp]),
        "si_engine_extract": (i, [vp, cp, C.POINTER(vp), C.POINTER(i)]),
        "si_engine_stream": (vp, [vp]),
        "si_engine_last_forward_ms": (C.c_float, [vp]),
        "si_engine_profile": (i, [vp]),
        "si_engine_profile_entry": (i, [vp, i, C.POINTER(cp), C.POINTER(cp), C.POINTER(cp), C.POINTER(C.c_float),
                                        C.POINTER(C.c_double), C.POINTER(C.c_double)]),
        "si_engine_schedule": (i, [vp, cp, sz]),
        "si_pnnx_dump": (i, [cp, cp, i, cp]),
        "si_pnnx_save": (i, [cp, cp, i, i, cp, cp]),
        "si_registry_types": (i, [cp, sz]),
    }
    # include/si_shard.h: node-local rank group + direct output all-gather
    shard = {
        "si_group_create": (i, [cp, i, i, C.c_double, C.POINTER(vp)]),
        "si_group_destroy": (i, [vp]),
        "si_group_rank": (i, [vp]),
        "si_group_world": (i, [vp]),
        "si_group_barrier": (i, [vp]),
        "si_group_allgather": (i, [vp, vp, sz, vp]),
        "si_gather_create": (i, [vp, i, sz, i, C.POINTER(vp)]),
        "si_gather_create_mode": (i, [vp, i, sz, i, i, C.POINTER(vp)]),
        "si_gather_mode": (i, [vp]),
        "si_rccl_available": (i, []),
        "si_rccl_init": (i, [vp, i, C.POINTER(vp)]),
        "si_rccl_allgather": (i, [vp, vp, vp, sz, vp]),
        "si_rccl_destroy": (i, [vp]),
        "si_gather_destroy": (i, [vp]),
        "si_gather_slots": (i, [vp]),
        "si_gather_slab_bytes": (sz, [vp]),
        "si_gather_buffer": (vp, [vp, i]),
        "si_gather_slab": (vp, [vp, i]),
        "si_gather_push": (i, [vp, i, vp]),
        "si_gather_complete": (i, [vp, i]),
        "si_gather_stats": (i, [vp, C.POINTER(SiGatherStats), i]),
    }
    for name, (res, args) in list(sig.items()) + list(shard.items()):
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = args
    L._si_signatures = sig
    L._si_shard_signatures = shard
    _host = L
    return L
