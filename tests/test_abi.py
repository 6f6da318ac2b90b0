"""CPU tests of the drop-in boundary: libogg_hip.so loads without a GPU and exports every symbol include/ogg_hip.h
declares; the ctypes signature table covers the header; argument errors surface with the reference's texts.  No
compute call is made here (there is no CPU compute path to call)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "ogg_hip.h")


def declared_symbols():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ogg_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_a_reasonable_surface():
    syms = declared_symbols()
    assert len(syms) >= 40
    for must in ("ogg_grid_metrics_midas", "ogg_angle_x", "ogg_bipolar_cap_metrics_quad", "ogg_displaced_pole_metrics_quad",
                 "ogg_bipolar_cap_mesh", "ogg_displaced_pole_mesh", "ogg_phi_mercator", "ogg_generate_latlon_grid"):
        assert must in syms


def test_library_exports_every_declared_symbol():
    from ocean_model_grid_generator_amd import _lib
    assert os.path.exists(_lib.LIB_PATH), "libogg_hip.so is not built: run __graft_entry__.build()"
    lib = ctypes.CDLL(_lib.LIB_PATH)
    missing = [s for s in declared_symbols() if not hasattr(lib, s)]
    assert not missing, missing


def test_ctypes_table_covers_the_header():
    from ocean_model_grid_generator_amd import _lib
    table = set(_lib.SIGNATURES) | set(_lib.STRING_GETTERS) | set(_lib.LONG_GETTERS)
    assert sorted(table) == declared_symbols()
    lib = _lib.load()
    assert lib.ogg_version().startswith(b"ogg_hip")


def test_descriptor_structs_have_the_c_layout():
    """The ctypes mirrors of the three descriptor structs must have the size the compiler gives them (a silent mismatch would
    shift every pointer), and the pass validates its arguments before any device work."""
    from ocean_model_grid_generator_amd import _lib
    lib = _lib.load()
    assert lib.ogg_abi_sizeof(0) == ctypes.sizeof(_lib.LatlonBand)
    assert lib.ogg_abi_sizeof(1) == ctypes.sizeof(_lib.BipolarBand)
    assert lib.ogg_abi_sizeof(2) == ctypes.sizeof(_lib.DpoleBand)
    assert lib.ogg_abi_sizeof(3) == -1
    bands = (_lib.LatlonBand * 5)()
    rc = lib.ogg_tripolar_pass_dev(5, bands, 1441, -300.0, 360.0, 6371e3, 1, None, None)     # more than 4 lat-lon bands
    assert rc == _lib.OGG_EARG and b"at most 4 bands" in lib.ogg_last_error()
    cap = _lib.BipolarBand()
    cap.Ni, cap.Nj, cap.n_pt_rows, cap.order = 1440, 10, 4, 9
    cap.x = cap.y = cap.angle = cap.dx = cap.dy = cap.area = 8     # never dereferenced: validation comes first
    rc = lib.ogg_tripolar_pass_dev(0, bands, 1441, -300.0, 360.0, 6371e3, 1, ctypes.byref(cap), None)
    assert rc == _lib.OGG_EORDER and lib.ogg_last_error() == b"Uncoded order"
    scap = _lib.DpoleBand()
    scap.Ni, scap.Nj, scap.n_pt_rows, scap.n_cell_rows, scap.order, scap.arc_form = 1440, 10, 4, 4, 3, 0
    scap.x = scap.y = scap.angle = scap.dx = scap.dy = scap.area = scap.workspace = 8
    scap.workspace_bytes = 1 << 30
    rc = lib.ogg_supergrid_pass_dev(0, bands, 1441, -300.0, 360.0, 6371e3, 1, None, ctypes.byref(scap), None, None, None)
    assert rc == _lib.OGG_EORDER and lib.ogg_last_error() == b"order not coded"        # OGG:547
    scap.order, scap.arc_form = 4, 7
    rc = lib.ogg_supergrid_pass_dev(0, bands, 1441, -300.0, 360.0, 6371e3, 1, None, ctypes.byref(scap), None, None, None)
    assert rc == _lib.OGG_EARG and b"arc_form" in lib.ogg_last_error()


def test_argument_errors_use_reference_texts():
    """Validation happens before any device work, so these run without a GPU."""
    from ocean_model_grid_generator_amd import _lib
    lib = _lib.load()
    rc = lib.ogg_bipolar_cap_metrics_quad(7, 48, 10, 64.0, -300.0, 0.2, 6371e3, None, None, None)
    assert rc == _lib.OGG_EORDER and lib.ogg_last_error() == b"Uncoded order"          # OGG:204
    rc = lib.ogg_displaced_pole_metrics_quad(3, 72, 14, -300.0, -78.0, 80.0, 0.2, 6371e3, None, None, None)
    assert rc == _lib.OGG_EORDER and lib.ogg_last_error() == b"order not coded"        # OGG:547
    with pytest.raises(Exception, match="Uncoded order"):
        _lib.check(lib.ogg_bipolar_cap_metrics_quad(1, 48, 10, 64.0, -300.0, 0.2, 6371e3, None, None, None))


def test_no_cpu_fallback_in_product_package():
    """The product package must not import the oracle (or anything under oracle/)."""
    pkg = os.path.join(ROOT, "ocean_model_grid_generator_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f
                assert "ogg_oracle" not in src, f
