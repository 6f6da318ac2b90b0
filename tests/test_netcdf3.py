"""CPU tests of the NetCDF-3 (64-bit offset) writer behind write_nc (OGG:773-829): its header and data encoding are pinned
byte for byte against scipy.io.netcdf_file(version=2) on the golden r0.25_even grid, and the streaming API (header first, data
written later at var_begin) produces the same file as write()."""
import os

import numpy as np

from ocean_model_grid_generator_amd import netcdf3

GOLD = os.path.join(os.path.dirname(__file__), "golden")
TILE = np.frombuffer(b"tile1".ljust(255, b"\0"), dtype="S1")


def _fields():
    g = np.load(os.path.join(GOLD, "ref_small_r0.25_even.npz"))
    return {k: g[k] for k in ("x", "y", "dx", "dy", "area", "angle_dx")}


def _spec(f, order):
    all_ = {"y": (("nyp", "nxp"), "degrees"), "x": (("nyp", "nxp"), "degrees"), "dy": (("ny", "nxp"), "meters"), "dx": (("nyp", "nx"), "meters"),
            "area": (("ny", "nx"), "m2"), "angle_dx": (("nyp", "nxp"), "degrees")}
    return [(n, all_[n][0], all_[n][1], f[n]) for n in order]


def _ours(path, f, order, gatts):
    ny, nx = f["area"].shape
    ds = netcdf3.Dataset(path, [("nyp", ny + 1), ("nxp", nx + 1), ("ny", ny), ("nx", nx), ("string", 255)], gatts)
    ds.def_var("tile", netcdf3.NC_CHAR, ("string",), [], TILE)
    for n, d, u, a in _spec(f, order):
        ds.def_var(n, netcdf3.NC_DOUBLE, d, [("units", u)], a)
    return ds


def test_bytes_equal_scipy(tmp_path):
    """scipy emits the non-record variables sorted by shape (descending), whatever the definition order; with our variables defined in
    that order the two files must be identical: magic, numrecs, dimension list, global and per-variable attributes, vsize, 64-bit
    begin offsets, padding and big-endian data."""
    from scipy.io import netcdf_file
    f = _fields()
    ny, nx = f["area"].shape
    order = ("y", "x", "angle_dx", "dx", "dy", "area")       # scipy's order for these shapes (tile, 255 > nyp, comes first)
    gatts = [("history", "made by a test"), ("description", "an orthogonal grid"), ("source", "s")]
    _ours(str(tmp_path / "ours.nc"), f, order, gatts).write()
    sp = netcdf_file(str(tmp_path / "scipy.nc"), "w", version=2)
    for k, v in gatts:
        setattr(sp, k, v)
    for n, l in (("nyp", ny + 1), ("nxp", nx + 1), ("ny", ny), ("nx", nx), ("string", 255)):
        sp.createDimension(n, l)
    v = sp.createVariable("tile", "c", ("string",))
    v[:] = TILE
    for n, d, u, a in _spec(f, order):
        v = sp.createVariable(n, "d", d)
        v.units = u
        v[:] = a
    sp.close()
    a, b = open(tmp_path / "ours.nc", "rb").read(), open(tmp_path / "scipy.nc", "rb").read()
    assert len(a) == len(b)
    assert a == b


def test_reference_layout_and_streaming_api(tmp_path):
    """write_nc's own order (tile, y, x, dy, dx, area, angle_dx; OGG:795-821): the file written in one go and the file written header
    first, variables later (as the device stream does) are identical, and scipy reads the fields back."""
    from scipy.io import netcdf_file
    f = _fields()
    order = ("y", "x", "dy", "dx", "area", "angle_dx")
    one = _ours(str(tmp_path / "one.nc"), f, order, [])
    one.write()
    ny, nx = f["area"].shape
    ds = netcdf3.Dataset(str(tmp_path / "two.nc"), [("nyp", ny + 1), ("nxp", nx + 1), ("ny", ny), ("nx", nx), ("string", 255)], [])
    ds.decl_var("tile", netcdf3.NC_CHAR, ("string",), [])
    for n, d, u, _ in _spec(f, order):
        ds.decl_var(n, netcdf3.NC_DOUBLE, d, [("units", u)])
    fd = os.open(str(tmp_path / "two.nc"), os.O_RDWR | os.O_CREAT | os.O_TRUNC, 0o644)
    ds.write_header(fd)
    os.pwrite(fd, TILE.tobytes(), ds.var_begin("tile"))
    for n in reversed(order):     # any order: every variable has its own byte range
        os.pwrite(fd, f[n].astype(">f8").tobytes(), ds.var_begin(n))
    os.close(fd)
    assert open(tmp_path / "one.nc", "rb").read() == open(tmp_path / "two.nc", "rb").read()
    assert one.layout() == ds.layout()[:1] + one.layout()[1:]      # same header bytes
    nc = netcdf_file(str(tmp_path / "two.nc"), "r", mmap=False)
    assert list(nc.dimensions.keys()) == ["nyp", "nxp", "ny", "nx", "string"]
    assert list(nc.variables.keys()) == ["tile", "y", "x", "dy", "dx", "area", "angle_dx"] and nc.version_byte == 2
    for n in order:
        assert np.array_equal(nc.variables[n][:], f[n])
    nc.close()
