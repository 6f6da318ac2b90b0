"""Minimal writer for the NetCDF classic format, 64-bit-offset variant (CDF-2) -- what the reference asks netCDF4 for
with format="NETCDF3_64BIT" (OGG:773-779).  Only what write_nc needs: fixed-size dimensions, char and double
variables, text attributes, variables laid out in definition order.  The file is a fixed header followed by the
variables' data, big-endian, each padded to 4 bytes, so writing it is a bandwidth-bound stream of the six fields.

Format reference: "The NetCDF Classic Format Specification" (header := magic numrecs dim_list gatt_list var_list).
"""
import struct

import numpy as np

NC_CHAR, NC_DOUBLE = 2, 6
NC_DIMENSION, NC_VARIABLE, NC_ATTRIBUTE = 0x0A, 0x0B, 0x0C
_ABSENT = struct.pack(">ii", 0, 0)


def _pad4(n):
    return (4 - n % 4) % 4


def _name(s):
    b = s.encode("utf-8")
    return struct.pack(">i", len(b)) + b + b"\0" * _pad4(len(b))


def _att_list(atts):
    if not atts:
        return _ABSENT
    out = struct.pack(">ii", NC_ATTRIBUTE, len(atts))
    for k, v in atts:
        b = v if isinstance(v, bytes) else str(v).encode("utf-8")
        out += _name(k) + struct.pack(">ii", NC_CHAR, len(b)) + b + b"\0" * _pad4(len(b))
    return out


class Dataset(object):
    """dims: [(name, length)], in order.  Variables are added with def_var (host array) or decl_var (layout only: the data is
    streamed into the file by the caller at var_begin(name)); write() writes header and host arrays, write_header() the header
    alone."""

    def __init__(self, path, dims, global_atts=()):
        self.path = path
        self.dims = list(dims)
        self.gatts = list(global_atts)
        self.vars = []  # (name, nc_type, dim names, attrs, array or None)
        self._layout = None

    def _shape(self, dim_names):
        return tuple(dict(self.dims)[d] for d in dim_names)

    def def_var(self, name, nc_type, dim_names, atts, data):
        shape = self._shape(dim_names)
        data = np.asarray(data)
        if tuple(data.shape) != shape:
            raise ValueError("variable %s: data shape %s does not match dimensions %s" % (name, data.shape, shape))
        self.vars.append((name, nc_type, tuple(dim_names), list(atts), data))
        self._layout = None

    def decl_var(self, name, nc_type, dim_names, atts):
        self.vars.append((name, nc_type, tuple(dim_names), list(atts), None))
        self._layout = None

    def layout(self):
        """(header bytes, [begin offset of each variable], [byte size of each variable], total file size)"""
        if self._layout is not None:
            return self._layout
        dimid = {n: k for k, (n, _) in enumerate(self.dims)}
        esize = {NC_CHAR: 1, NC_DOUBLE: 8}

        def var_header(name, nc_type, dnames, atts, nbytes, begin):
            vsize = nbytes + _pad4(nbytes)
            if vsize > 2 ** 32 - 4:
                vsize = 2 ** 32 - 1
            h = _name(name) + struct.pack(">i", len(dnames)) + b"".join(struct.pack(">i", dimid[d]) for d in dnames)
            return h + _att_list(atts) + struct.pack(">iI", nc_type, vsize) + struct.pack(">q", begin)

        head = b"CDF\x02" + struct.pack(">i", 0)
        head += struct.pack(">ii", NC_DIMENSION, len(self.dims)) + b"".join(_name(n) + struct.pack(">i", l) for n, l in self.dims)
        head += _att_list(self.gatts)
        sizes = [int(np.prod(self._shape(v[2]), dtype=np.int64)) * esize[v[1]] for v in self.vars]
        # header length does not depend on the begin values (fixed-width), so compute it with zeros first
        var_list = struct.pack(">ii", NC_VARIABLE, len(self.vars)) if self.vars else _ABSENT
        probe = head + var_list + b"".join(var_header(v[0], v[1], v[2], v[3], s, 0) for v, s in zip(self.vars, sizes))
        begin = len(probe)
        body = b""
        begins = []
        for v, s in zip(self.vars, sizes):
            begins.append(begin)
            body += var_header(v[0], v[1], v[2], v[3], s, begin)
            begin += s + _pad4(s)
        self._layout = (head + var_list + body, begins, sizes, begin)
        return self._layout

    def var_begin(self, name):
        _, begins, _, _ = self.layout()
        return begins[[v[0] for v in self.vars].index(name)]

    def write_header(self, fd):
        """Header at offset 0 of the open file descriptor, file extended to its final size (the padding bytes are zeros)."""
        import os
        header, _, _, total = self.layout()
        os.ftruncate(fd, total)
        os.pwrite(fd, header, 0)

    def write(self, chunk_rows=256):
        header, begins, sizes, _ = self.layout()
        with open(self.path, "wb") as f:
            f.write(header)
            for v, s, b in zip(self.vars, sizes, begins):
                assert f.tell() == b
                data = v[4]
                if data is None:
                    raise ValueError("variable %s was declared without data: use write_header() and stream it" % v[0])
                if v[1] == NC_CHAR:
                    f.write(np.ascontiguousarray(data).tobytes())
                else:
                    flat = data.reshape(-1, data.shape[-1]) if data.ndim > 1 else data.reshape(1, -1)
                    for r0 in range(0, flat.shape[0], chunk_rows):
                        f.write(np.ascontiguousarray(flat[r0:r0 + chunk_rows]).astype(">f8").tobytes())
                f.write(b"\0" * _pad4(s))
