"""ctypes binding of libogg_hip.so (the C ABI declared in include/ogg_hip.h).

There is no fallback: if the shared library cannot be loaded, or a call returns an error code, an exception is
raised.  Error codes map to the reference's own exception texts where it has them (OGG:204, OGG:547, OGG:722).
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# OGG_LIB_PATH: another build of the same library (A/B timing of two builds on one box, scripts/ab_libs.sh); it must export the same ABI
LIB_PATH = os.environ.get("OGG_LIB_PATH") or os.path.join(_HERE, "csrc", "libogg_hip.so")

OGG_OK, OGG_EORDER, OGG_ESHAPE, OGG_EHIP, OGG_ENOMEM, OGG_EARG = 0, 1, 2, 3, 4, 5
DP_ARC_LITERAL, DP_ARC_CHORD = 0, 1   # OGG_DP_ARC_* of include/ogg_hip.h
SYM_DEFAULT, SYM_MIRROR, SYM_NONE = 0, 1, 2   # OGG_SYM_* of include/ogg_hip.h

c_long, c_int, c_double, c_void_p, c_longlong = ctypes.c_long, ctypes.c_int, ctypes.c_double, ctypes.c_void_p, ctypes.c_longlong
_dp = ctypes.POINTER(ctypes.c_double)
_llp = ctypes.POINTER(ctypes.c_longlong)

class LatlonBand(ctypes.Structure):
    """ogg_latlon_band of include/ogg_hip.h"""
    _fields_ = [("axis_kind", c_int), ("a0", c_double), ("len", c_double), ("denom", c_double), ("y0", c_longlong),
                ("lat1d", c_void_p), ("k0", c_long), ("n_pt_rows", c_long), ("n_cell_rows", c_long), ("x", c_void_p),
                ("y", c_void_p), ("dx", c_void_p), ("dy", c_void_p), ("area", c_void_p), ("angle", c_void_p)]


class BipolarBand(ctypes.Structure):
    """ogg_bipolar_band of include/ogg_hip.h"""
    _fields_ = [("Ni", c_long), ("Nj", c_long), ("lat0_bp", c_double), ("lon_bp", c_double), ("rp", c_double), ("Re", c_double),
                ("order", c_int), ("symmetry", c_int), ("j0", c_long), ("n_pt_rows", c_long), ("n_cell_rows", c_long), ("x", c_void_p),
                ("y", c_void_p), ("angle", c_void_p), ("dx", c_void_p), ("dy", c_void_p), ("area", c_void_p), ("workspace", c_void_p),
                ("workspace_bytes", c_long)]


class DpoleBand(ctypes.Structure):
    """ogg_dpole_band of include/ogg_hip.h"""
    _fields_ = [("Ni", c_long), ("Nj", c_long), ("lon0", c_double), ("lat0", c_double), ("lon_dp", c_double), ("r_dp", c_double),
                ("Re", c_double), ("order", c_int), ("arc_form", c_int), ("j0", c_long), ("n_pt_rows", c_long), ("n_cell_rows", c_long),
                ("x", c_void_p), ("y", c_void_p), ("angle", c_void_p), ("dx", c_void_p), ("dy", c_void_p), ("area", c_void_p),
                ("workspace", c_void_p), ("workspace_bytes", c_long), ("symmetry", c_int)]


# name -> argtypes; every function returns int except the two string getters.  Must list EVERY symbol of ogg_hip.h
# (tests/test_abi.py checks this list against the header).
SIGNATURES = {
    "ogg_device_count": [ctypes.POINTER(c_int)],
    "ogg_set_device": [c_int],
    "ogg_device_name": [ctypes.c_char_p, c_int],
    "ogg_y_mercator_rounded": [c_long, c_long, c_void_p, c_void_p],
    "ogg_y_mercator_rounded_dev": [c_long, c_long, c_void_p, c_void_p, c_void_p],
    "ogg_phi_mercator": [c_long, c_long, c_void_p, c_void_p],
    "ogg_phi_mercator_dev": [c_long, c_long, c_void_p, c_void_p, c_void_p],
    "ogg_mercator_axis_dev": [c_long, c_longlong, c_long, c_void_p, c_void_p],
    "ogg_linear_axis_dev": [c_long, c_double, c_double, c_double, c_void_p, c_void_p],
    "ogg_tile_latlon": [c_long, c_long, c_void_p, c_void_p, c_void_p, c_void_p],
    "ogg_tile_latlon_dev": [c_long, c_long, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p],
    "ogg_generate_latlon_grid": [c_long, c_long, c_double, c_double, c_double, c_double, c_int, c_void_p, c_void_p],
    "ogg_grid_metrics_midas_dev": [c_long, c_long, c_void_p, c_void_p, c_long, c_long, c_double, c_int, c_void_p, c_void_p,
                                   c_void_p, c_void_p, c_void_p],
    "ogg_grid_metrics_midas": [c_long, c_long, c_void_p, c_void_p, c_double, c_int, c_void_p, c_void_p, c_void_p],
    "ogg_angle_x": [c_long, c_long, c_void_p, c_void_p, c_void_p],
    "ogg_bipolar_projection": [c_long, c_void_p, c_void_p, c_double, c_double, c_int, c_void_p, c_void_p, c_void_p, c_void_p],
    "ogg_bipolar_projection_dev": [c_long, c_void_p, c_void_p, c_double, c_double, c_int, c_void_p, c_void_p, c_void_p,
                                   c_void_p, c_void_p],
    "ogg_bipolar_cap_mesh_dev": [c_long, c_long, c_double, c_double, c_long, c_long, c_void_p, c_void_p, c_void_p, c_void_p,
                                 c_void_p],
    "ogg_bipolar_cap_mesh_angle_dev": [c_long, c_long, c_double, c_double, c_long, c_long, c_void_p, c_void_p, c_void_p, c_void_p,
                                       c_void_p, c_void_p],
    "ogg_bipolar_cap_mesh_sym": [c_long, c_long, c_double, c_double, c_int, c_void_p, c_void_p, c_void_p, c_void_p],
    "ogg_bipolar_cap_metrics_quad_sym": [c_int, c_long, c_long, c_double, c_double, c_double, c_double, c_int, c_void_p, c_void_p, c_void_p],
    "ogg_displaced_pole_metrics_quad_form_sym": [c_int, c_int, c_int, c_long, c_long, c_double, c_double, c_double, c_double, c_double, c_void_p,
                                                 c_void_p, c_void_p],
    "ogg_symmetry_coverage": [c_int, c_int, c_long, c_double, c_double, c_int, c_void_p, c_void_p, c_void_p],
    "ogg_bipolar_cap_mesh_angle_sym_dev": [c_long, c_long, c_double, c_double, c_long, c_long, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                                           c_void_p, c_void_p],
    "ogg_bipolar_cap_mesh": [c_long, c_long, c_double, c_double, c_void_p, c_void_p, c_void_p, c_void_p],
    "ogg_bipolar_cap_metrics_quad_dev": [c_int, c_long, c_long, c_double, c_double, c_double, c_double, c_long, c_long, c_long,
                                         c_void_p, c_void_p, c_void_p, c_void_p],
    "ogg_bipolar_cap_metrics_quad_ws_dev": [c_int, c_long, c_long, c_double, c_double, c_double, c_double, c_long, c_long, c_long,
                                            c_void_p, c_void_p, c_void_p, c_void_p, c_long, c_void_p],
    "ogg_bipolar_cap_metrics_quad_sym_ws_dev": [c_int, c_long, c_long, c_double, c_double, c_double, c_double, c_long, c_long, c_long, c_int,
                                                c_void_p, c_void_p, c_void_p, c_void_p, c_long, c_void_p],
    "ogg_bipolar_cap_metrics_quad": [c_int, c_long, c_long, c_double, c_double, c_double, c_double, c_void_p, c_void_p,
                                     c_void_p],
    "ogg_displaced_pole_mesh_dev": [c_long, c_void_p, c_long, c_void_p, c_long, c_long, c_double, c_double, c_double,
                                    c_double, c_void_p, c_void_p, c_void_p],
    "ogg_displaced_pole_mesh": [c_long, c_void_p, c_long, c_void_p, c_long, c_long, c_double, c_double, c_double, c_double,
                                c_void_p, c_void_p],
    "ogg_displaced_pole_grid_dev": [c_long, c_long, c_double, c_double, c_double, c_double, c_long, c_long, c_void_p,
                                    c_void_p, c_void_p],
    "ogg_displaced_pole_numerical_h_dev": [c_long, c_void_p, c_long, c_void_p, c_long, c_long, c_double, c_double, c_double,
                                           c_double, c_double, c_int, c_void_p, c_void_p, c_void_p],
    "ogg_displaced_pole_numerical_h": [c_long, c_void_p, c_long, c_void_p, c_long, c_long, c_double, c_double, c_double,
                                       c_double, c_double, c_int, c_void_p, c_void_p],
    "ogg_displaced_pole_metrics_quad_dev": [c_int, c_long, c_long, c_double, c_double, c_double, c_double, c_double, c_long,
                                            c_long, c_long, c_void_p, c_void_p, c_void_p, c_void_p],
    "ogg_displaced_pole_metrics_quad_ws_dev": [c_int, c_long, c_long, c_double, c_double, c_double, c_double, c_double, c_long,
                                               c_long, c_long, c_void_p, c_void_p, c_void_p, c_void_p, c_long, c_void_p],
    "ogg_displaced_pole_metrics_quad": [c_int, c_long, c_long, c_double, c_double, c_double, c_double, c_double, c_void_p,
                                        c_void_p, c_void_p],
    "ogg_displaced_pole_metrics_quad_form": [c_int, c_int, c_long, c_long, c_double, c_double, c_double, c_double, c_double, c_void_p,
                                             c_void_p, c_void_p],
    "ogg_displaced_pole_metrics_quad_form_ws_dev": [c_int, c_int, c_long, c_long, c_double, c_double, c_double, c_double, c_double,
                                                    c_long, c_long, c_long, c_void_p, c_void_p, c_void_p, c_void_p, c_long, c_void_p],
    "ogg_displaced_pole_metrics_quad_form_sym_ws_dev": [c_int, c_int, c_int, c_long, c_long, c_double, c_double, c_double, c_double, c_double,
                                                        c_long, c_long, c_long, c_void_p, c_void_p, c_void_p, c_void_p, c_long, c_void_p],
    "ogg_displaced_pole_grid_angle_ws_dev": [c_long, c_long, c_double, c_double, c_double, c_double, c_long, c_long, c_void_p,
                                             c_void_p, c_void_p, c_void_p, c_long, c_void_p],
    "ogg_workspace_error_flag_dev": [c_void_p, ctypes.POINTER(c_int), c_void_p],
    "ogg_y_mercator": [c_long, c_long, c_void_p, c_void_p],
    "ogg_y_mercator_dev": [c_long, c_long, c_void_p, c_void_p, c_void_p],
    "ogg_affine_index": [c_long, c_void_p, c_double, c_double, c_double, c_void_p],
    "ogg_affine_index_dev": [c_long, c_void_p, c_double, c_double, c_double, c_void_p, c_void_p],
    "ogg_mdist": [c_long, c_void_p, c_void_p, c_void_p],
    "ogg_mdist_dev": [c_long, c_void_p, c_void_p, c_void_p, c_void_p],
    "ogg_haversine": [c_long, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p],
    "ogg_haversine_dev": [c_long, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p],
    "ogg_bipolar_cap_ij_array": [c_long, c_void_p, c_long, c_void_p, c_long, c_long, c_double, c_double, c_double, c_void_p,
                                 c_void_p],
    "ogg_bipolar_cap_ij_array_dev": [c_long, c_void_p, c_long, c_void_p, c_long, c_long, c_double, c_double, c_double,
                                     c_void_p, c_void_p, c_void_p],
    "ogg_displaced_pole_projection": [c_long, c_long, c_void_p, c_void_p, c_double, c_double, c_double, c_double, c_void_p,
                                      c_void_p],
    "ogg_displaced_pole_projection_dev": [c_long, c_long, c_void_p, c_void_p, c_double, c_double, c_double, c_double,
                                          c_void_p, c_void_p, c_void_p],
    "ogg_monotonic_bounding": [c_long, c_long, c_void_p, c_double],
    "ogg_monotonic_bounding_dev": [c_long, c_long, c_void_p, c_double, c_void_p],
    "ogg_latlon_supergrid_multi_dev": [c_int, ctypes.POINTER(LatlonBand), c_long, c_double, c_double, c_double, c_int, c_void_p],
    "ogg_latlon_supergrid_rows_ws_dev": [c_int, ctypes.POINTER(LatlonBand), c_long, c_double, c_double, c_double, c_int, c_void_p, c_long,
                                         c_void_p],
    "ogg_tripolar_pass_dev": [c_int, ctypes.POINTER(LatlonBand), c_long, c_double, c_double, c_double, c_int,
                              ctypes.POINTER(BipolarBand), c_void_p],
    "ogg_tripolar_pass_events_dev": [c_int, ctypes.POINTER(LatlonBand), c_long, c_double, c_double, c_double, c_int,
                                     ctypes.POINTER(BipolarBand), ctypes.POINTER(c_void_p), ctypes.POINTER(c_double), c_void_p],
    "ogg_supergrid_pass_dev": [c_int, ctypes.POINTER(LatlonBand), c_long, c_double, c_double, c_double, c_int,
                               ctypes.POINTER(BipolarBand), ctypes.POINTER(DpoleBand), ctypes.POINTER(c_void_p), ctypes.POINTER(c_double),
                               c_void_p],
    "ogg_supergrid_pass_plan_dev": [c_int, ctypes.POINTER(LatlonBand), c_long, c_double, c_double, c_double, c_int,
                                    ctypes.POINTER(BipolarBand), ctypes.POINTER(DpoleBand), ctypes.POINTER(c_void_p)],
    "ogg_supergrid_pass_run_dev": [c_void_p, ctypes.POINTER(c_void_p), ctypes.POINTER(c_double), c_void_p],
    "ogg_supergrid_pass_plan_destroy": [c_void_p],
    "ogg_supergrid_pass_plan_flags_dev": [c_void_p, ctypes.POINTER(c_int), c_void_p],
    "ogg_latlon_supergrid_dev": [c_long, c_long, c_long, c_void_p, c_void_p, c_double, c_int, c_void_p, c_void_p, c_void_p,
                                 c_void_p, c_void_p, c_void_p, c_void_p],
    "ogg_fill_dev": [c_long, c_double, c_void_p, c_void_p],
    "ogg_bswap64_dev": [c_long, c_void_p, c_void_p, c_void_p],
    "ogg_libm_check_dev": [c_int, c_long, c_void_p, c_void_p, c_void_p, c_void_p],
    "ogg_metrics_sums_dev": [c_long, c_long, c_long, c_void_p, c_void_p, c_void_p, c_long, c_long, c_int, c_int, c_void_p, c_void_p],
    "ogg_event_create": [ctypes.POINTER(c_void_p)],
    "ogg_event_destroy": [c_void_p],
    "ogg_event_record": [c_void_p, c_void_p],
    "ogg_event_elapsed_ms": [c_void_p, c_void_p, ctypes.POINTER(ctypes.c_float)],
    "ogg_stream_synchronize": [c_void_p],
}
STRING_GETTERS = ("ogg_last_error", "ogg_version")
LONG_GETTERS = {"ogg_abi_sizeof": [c_int],
                "ogg_bipolar_quad_workspace_bytes": [c_int, c_long, c_long],
                "ogg_displaced_pole_quad_workspace_bytes": [c_int, c_long, c_long],
                "ogg_displaced_pole_grid_workspace_bytes": [c_long, c_long],
                "ogg_dpole_band_workspace_bytes": [c_int, c_long, c_long],
                "ogg_latlon_rows_workspace_bytes": [c_int, ctypes.POINTER(LatlonBand), c_long],
                "ogg_supergrid_pass_plan_slots": [c_void_p],
                "ogg_supergrid_pass_plan_carried_runs": [c_void_p]}

_lib = None


class OggHipError(Exception):
    """A libogg_hip.so call failed; .code is the OGG_E* value."""

    def __init__(self, code, text):
        Exception.__init__(self, text)
        self.code = code


def load():
    """Load libogg_hip.so (once).  Raises if it is missing: there is no CPU path behind this package."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError("libogg_hip.so is not built (%s). Run `python -m ocean_model_grid_generator_amd.csrc.build` "
                          "(needs hipcc); this package has no CPU fallback." % LIB_PATH)
    # One HIP runtime per process: the PyTorch wheel bundles its own libamdhip64 / libhsa-runtime64, and whichever
    # runtime touches the GPU second sees no device.  Importing torch first makes libogg_hip.so bind to the runtime
    # torch already loaded (same SONAME), so torch tensors, streams and RCCL share one runtime with our kernels.
    # Without torch installed the system runtime under /opt/rocm is used.  OGG_NO_TORCH=1 skips the import.
    if not os.environ.get("OGG_NO_TORCH"):
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
    lib = ctypes.CDLL(LIB_PATH)
    for name, argtypes in SIGNATURES.items():
        if os.environ.get("OGG_LIB_PATH") and not hasattr(lib, name):
            continue   # an older build under A/B timing (scripts/ab_time.py): the entry points it lacks are not called there
        fn = getattr(lib, name)
        fn.argtypes = argtypes
        fn.restype = c_int
    for name in STRING_GETTERS:
        getattr(lib, name).restype = ctypes.c_char_p
        getattr(lib, name).argtypes = []
    for name, argtypes in LONG_GETTERS.items():
        getattr(lib, name).restype = c_long
        getattr(lib, name).argtypes = argtypes
    _lib = lib
    return lib


def check(code):
    """Turn a non-zero return code into the exception the reference would raise."""
    if code == OGG_OK:
        return
    text = load().ogg_last_error().decode("utf-8", "replace")
    if code in (OGG_EORDER, OGG_ESHAPE):
        # reference texts: "Uncoded order", "order not coded", "Input arrays do not have the same shape!"
        raise Exception(text)
    raise OggHipError(code, "libogg_hip: %s" % text)


def call(name, *args):
    check(getattr(load(), name)(*args))


def ptr(a):
    """Host pointer of a C-contiguous float64/int64 numpy array (or None)."""
    if a is None:
        return None
    assert a.flags["C_CONTIGUOUS"]
    return a.ctypes.data


def as_f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def source_hash():
    """The kernel-source hash compiled into the loaded library (the tail of ogg_version())."""
    v = load().ogg_version().decode()
    return v.split(" src ")[1] if " src " in v else None


def device_count():
    n = c_int(0)
    rc = load().ogg_device_count(ctypes.byref(n))
    return n.value if rc == OGG_OK else 0


def device_name():
    buf = ctypes.create_string_buffer(256)
    call("ogg_device_name", buf, 256)
    return buf.value.decode()
