"""ctypes binding of libplaysnark_hip.so.  Fails loudly: no library => ImportError, there is
no Python/CPU implementation behind these calls."""
from __future__ import annotations

import ctypes as C
import os

from .build import library_path

PS_OK = 0
PS_ERR_LENGTH = -1
PS_ERR_NOT_DIVISIBLE = -2
PS_ERR_ENCODING = -3
PS_ERR_HIP = -4
PS_ERR_ARG = -5
PS_ERR_NO_DEVICE = -6
PS_FMT_AFFINE, PS_FMT_COMPRESSED = 0, 1
PS_G1, PS_G2 = 1, 2

# every symbol include/playsnark_hip.h declares (tests/test_abi.py checks the header against this)
SYMBOLS = [
    "ps_abi_version", "ps_last_error", "ps_version", "ps_device_count",
    "ps_ctx_create", "ps_ctx_destroy", "ps_ctx_sync", "ps_ctx_stream", "ps_ctx_set_tables", "ps_ctx_set_table_budget",
    "ps_points_upload", "ps_points_from_scalars", "ps_points_download", "ps_points_download_fmt", "ps_points_len", "ps_points_group",
    "ps_points_slice", "ps_points_free", "ps_points_check_subgroup", "ps_points_precompute", "ps_points_table_window",
    "ps_scalars_upload", "ps_scalars_upload_i64", "ps_scalars_from_device_be32", "ps_scalars_download",
    "ps_scalars_len", "ps_scalars_slice", "ps_scalars_free",
    "ps_msm", "ps_msm_be32", "ps_msm_i64", "ps_msm_launch", "ps_msm_finish", "ps_msm_multi", "ps_points_sum", "ps_point_convert",
    "ps_msm_last_info", "ps_msm_set_window", "ps_msm_set_slice", "ps_msm_set_tail", "ps_microbench_mad", "ps_ctx_set_timing", "ps_msm_last_stage_ms",
    "ps_qap_create", "ps_qap_free", "ps_qap_quotient", "ps_qap_is_valid", "ps_qap_interpolate", "ps_poly_mul",
    "ps_points_lincomb", "ps_msm_multi_device", "ps_groth16_prove_multi", "ps_points_monomial_to_lagrange",
    "ps_groth16_setup", "ps_phgr13_setup", "ps_phgr13_crs_free", "ps_groth16_prove", "ps_groth16_prove_shard", "ps_phgr13_prove", "ps_groth16_verify", "ps_phgr13_verify", "ps_pairing_equal", "ps_prove_last_phase_ms",
]


PS_MSM_QUEUE = 4  # pending sums per context (include/playsnark_hip.h)


class MsmInfo(C.Structure):
    _fields_ = [("window_bits", C.c_int), ("windows", C.c_int), ("entries", C.c_uint64),
                ("buckets", C.c_uint64), ("slice", C.c_int), ("window_table", C.c_int)]


class Csr(C.Structure):
    _fields_ = [("row_ptr", C.c_void_p), ("col", C.c_void_p), ("val", C.c_void_p)]


class Groth16Pk(C.Structure):
    _fields_ = [("alpha", C.c_uint8 * 96), ("beta", C.c_uint8 * 96), ("delta", C.c_uint8 * 96),
                ("beta2", C.c_uint8 * 192), ("delta2", C.c_uint8 * 192),
                ("xi", C.c_void_p), ("xi2", C.c_void_p), ("nio_lp", C.c_void_p), ("xi_t", C.c_void_p),
                ("lxi", C.c_void_p), ("lxi2", C.c_void_p), ("lxi_t", C.c_void_p)]  # the Lagrange form, NULL = absent


class Groth16Device(C.Structure):
    _fields_ = [("ctx", C.c_void_p), ("qap", C.c_void_p), ("sol", C.c_void_p), ("pk", Groth16Pk)]


class Groth16Toxic(C.Structure):
    _fields_ = [(n, C.c_uint8 * 32) for n in ("alpha", "beta", "delta", "x", "gamma")]


class Groth16Crs(C.Structure):
    _fields_ = [("alpha", C.c_uint8 * 96), ("beta", C.c_uint8 * 96), ("delta", C.c_uint8 * 96),
                ("beta2", C.c_uint8 * 192), ("delta2", C.c_uint8 * 192), ("gamma", C.c_uint8 * 192),
                ("xi", C.c_void_p), ("xi2", C.c_void_p), ("io_lp", C.c_void_p), ("nio_lp", C.c_void_p), ("xi_t", C.c_void_p),
                ("lxi", C.c_void_p), ("lxi2", C.c_void_p), ("lxi_t", C.c_void_p)]


class Groth16Vk(C.Structure):
    _fields_ = [("alpha", C.c_uint8 * 96), ("beta2", C.c_uint8 * 192), ("gamma", C.c_uint8 * 192),
                ("delta2", C.c_uint8 * 192), ("io_lp", C.c_void_p)]


class Phgr13Vk(C.Structure):
    _fields_ = [("av", C.c_uint8 * 192), ("aw", C.c_uint8 * 96), ("ay", C.c_uint8 * 192), ("gamma", C.c_uint8 * 192),
                ("bgamma", C.c_uint8 * 96), ("bgamma2", C.c_uint8 * 192), ("yts", C.c_uint8 * 192),
                ("vs_io", C.c_void_p), ("ws_io", C.c_void_p), ("ys_io", C.c_void_p)]


class Phgr13Ek(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("vs", "ws", "ys", "vas", "was", "yas", "gsi", "vbs", "wbs", "ybs", "lgsi")]


class Phgr13Toxic(C.Structure):
    _fields_ = [(n, C.c_uint8 * 32) for n in ("s", "av", "aw", "ay", "rv", "rw", "beta", "gamma")]


class Phgr13Crs(C.Structure):
    _fields_ = ([(n, C.c_void_p) for n in ("gsi", "vs", "ws", "ys", "vas", "was", "yas", "vbs", "wbs", "ybs")] +
                [("av", C.c_uint8 * 192), ("aw", C.c_uint8 * 96), ("ay", C.c_uint8 * 192), ("gamma", C.c_uint8 * 192),
                 ("bgamma", C.c_uint8 * 96), ("bgamma2", C.c_uint8 * 192), ("yts", C.c_uint8 * 192)] +
                [(n, C.c_void_p) for n in ("vk_vs", "vk_ws", "vk_ys", "lgsi")])


class Phgr13Proof(C.Structure):
    _fields_ = [("vss", C.c_uint8 * 96), ("vass", C.c_uint8 * 96), ("wss", C.c_uint8 * 192),
                ("wass", C.c_uint8 * 96), ("yss", C.c_uint8 * 96), ("yass", C.c_uint8 * 96),
                ("hs", C.c_uint8 * 96), ("gz", C.c_uint8 * 96)]


def _load():
    path = library_path()
    if not os.path.exists(path):
        raise ImportError(
            f"{path} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  playsnark_amd has no CPU fallback."
        )
    lib = C.CDLL(path)
    lib.ps_last_error.restype = C.c_char_p
    lib.ps_version.restype = C.c_char_p
    lib.ps_ctx_stream.restype = C.c_void_p
    lib.ps_points_len.restype = C.c_size_t
    lib.ps_scalars_len.restype = C.c_size_t
    vp, sz, i = C.c_void_p, C.c_size_t, C.c_int
    pp = C.POINTER(C.c_void_p)
    lib.ps_ctx_create.argtypes = [i, pp]
    lib.ps_ctx_destroy.argtypes = [vp]
    lib.ps_ctx_destroy.restype = None
    lib.ps_ctx_sync.argtypes = [vp]
    lib.ps_ctx_stream.argtypes = [vp]
    lib.ps_ctx_set_tables.argtypes = [vp, i]
    lib.ps_points_upload.argtypes = [vp, i, C.c_char_p, sz, i, pp]
    lib.ps_points_from_scalars.argtypes = [vp, i, vp, pp]
    lib.ps_points_download.argtypes = [vp, vp, sz, sz, C.c_char_p]
    lib.ps_points_download_fmt.argtypes = [vp, vp, sz, sz, i, C.c_char_p]
    lib.ps_points_len.argtypes = [vp]
    lib.ps_points_group.argtypes = [vp]
    lib.ps_points_slice.argtypes = [vp, sz, sz, pp]
    lib.ps_points_check_subgroup.argtypes = [vp, vp, C.POINTER(C.c_int)]
    lib.ps_points_precompute.argtypes = [vp, vp, i]
    lib.ps_points_table_window.argtypes = [vp]
    lib.ps_points_free.argtypes = [vp]
    lib.ps_points_free.restype = None
    lib.ps_scalars_upload.argtypes = [vp, C.c_char_p, sz, pp]
    lib.ps_scalars_upload_i64.argtypes = [vp, vp, sz, pp]
    lib.ps_scalars_from_device_be32.argtypes = [vp, vp, sz, pp]
    lib.ps_scalars_download.argtypes = [vp, vp, sz, sz, C.c_char_p]
    lib.ps_scalars_len.argtypes = [vp]
    lib.ps_scalars_slice.argtypes = [vp, sz, sz, pp]
    lib.ps_scalars_free.argtypes = [vp]
    lib.ps_scalars_free.restype = None
    lib.ps_msm.argtypes = [vp, vp, vp, C.c_char_p]
    lib.ps_msm_be32.argtypes = [vp, vp, C.c_char_p, sz, C.c_char_p]
    lib.ps_msm_i64.argtypes = [vp, vp, vp, sz, C.c_char_p]
    lib.ps_msm_launch.argtypes = [vp, vp, vp]
    lib.ps_msm_finish.argtypes = [vp, C.c_char_p]
    lib.ps_msm_multi.argtypes = [vp, C.POINTER(vp), C.c_size_t, vp, C.POINTER(vp)]
    lib.ps_points_sum.argtypes = [i, C.c_char_p, sz, C.c_char_p]
    lib.ps_point_convert.argtypes = [i, i, i, C.c_char_p, C.c_char_p]
    lib.ps_msm_last_info.argtypes = [vp, C.POINTER(MsmInfo)]
    lib.ps_msm_set_window.argtypes = [vp, i]
    lib.ps_ctx_set_timing.argtypes = [vp, i]
    lib.ps_msm_set_slice.argtypes = [vp, i]
    lib.ps_msm_set_tail.argtypes = [vp, i]
    lib.ps_microbench_mad.argtypes = [vp, C.POINTER(C.c_double)]
    lib.ps_ctx_set_table_budget.argtypes = [vp, C.c_longlong]
    lib.ps_qap_is_valid.argtypes = [vp, vp, vp, C.POINTER(C.c_int)]
    lib.ps_msm_last_stage_ms.argtypes = [vp, C.POINTER(C.c_float)]
    lib.ps_qap_create.argtypes = [vp, sz, sz, sz, C.POINTER(Csr), C.POINTER(Csr), C.POINTER(Csr), pp]
    lib.ps_qap_free.argtypes = [vp]
    lib.ps_qap_free.restype = None
    lib.ps_qap_quotient.argtypes = [vp, vp, vp, pp, pp, pp, pp]
    lib.ps_poly_mul.argtypes = [vp, vp, vp, pp]
    lib.ps_qap_interpolate.argtypes = [vp, vp, vp, i, pp]
    lib.ps_points_lincomb.argtypes = [i, C.c_char_p, C.c_char_p, sz, C.c_char_p]
    lib.ps_points_monomial_to_lagrange.argtypes = [vp, vp, vp, i, C.POINTER(vp)]
    lib.ps_msm_multi_device.argtypes = [C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), sz, C.c_char_p]
    lib.ps_groth16_prove_multi.argtypes = [C.POINTER(Groth16Device), sz, C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p]
    lib.ps_groth16_prove.argtypes = [vp, C.POINTER(Groth16Pk), vp, vp, C.c_char_p, C.c_char_p, C.c_char_p,
                                     C.c_char_p, C.c_char_p]
    lib.ps_groth16_prove_shard.argtypes = [vp, C.POINTER(Groth16Pk), vp, vp, C.c_char_p, C.c_char_p, i, i, C.c_char_p,
                                           C.c_char_p, C.c_char_p]
    lib.ps_phgr13_prove.argtypes = [vp, C.POINTER(Phgr13Ek), vp, vp, C.POINTER(Phgr13Proof)]
    lib.ps_groth16_setup.argtypes = [vp, vp, C.POINTER(Groth16Toxic), C.POINTER(Groth16Crs)]
    lib.ps_phgr13_setup.argtypes = [vp, vp, C.POINTER(Phgr13Toxic), C.POINTER(Phgr13Crs)]
    lib.ps_phgr13_crs_free.argtypes = [C.POINTER(Phgr13Crs)]
    lib.ps_phgr13_crs_free.restype = None
    lib.ps_prove_last_phase_ms.argtypes = [vp, C.POINTER(C.c_float)]
    lib.ps_groth16_verify.argtypes = [vp, C.POINTER(Groth16Vk), vp, C.c_char_p, C.c_char_p, C.c_char_p, C.POINTER(C.c_int)]
    lib.ps_phgr13_verify.argtypes = [vp, C.POINTER(Phgr13Vk), vp, C.POINTER(Phgr13Proof), C.POINTER(C.c_int)]
    lib.ps_pairing_equal.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.POINTER(C.c_int)]
    return lib


lib = _load()
if lib.ps_abi_version() != 4:  # include/playsnark_hip.h PS_ABI_VERSION: the structs above mirror that revision
    raise ImportError("libplaysnark_hip.so has ABI %d, this binding is written for 3" % lib.ps_abi_version())
