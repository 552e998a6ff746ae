"""Minimal HDF5 reader for the two netCDF4 files the reference ships next to its masking notebook
(examples/ngwerere/ngwerere_piv.nc -> ngwerere_masked.nc): TEST INFRASTRUCTURE ONLY, used by tests/golden/make_golden.py
in the build container (h5py / netCDF4 / xarray are not installable here) to turn those files into .npz fixtures.

Reads exactly what those files contain: superblock v2, version-2 object headers with continuation blocks, chunked
layout v3 indexed by a version-1 B-tree, the shuffle + deflate (+ fletcher32) filter pipeline, fixed-point and
floating-point datatypes.  Datasets are located by their hard link (name followed by the object-header address, the
tail of every HDF5 link message), which avoids walking the dense link storage (fractal heap + v2 B-tree) of the root group.
Follows the published HDF5 File Format Specification, version 3.0; no reference code involved.
"""
import struct
import zlib

import numpy as np

UNDEF = 0xFFFFFFFFFFFFFFFF


def _u(buf, off, n):
    return int.from_bytes(buf[off:off + n], "little")


def object_headers(buf):
    """Offsets of all version-2 object headers."""
    out, i = [], buf.find(b"OHDR")
    while i >= 0:
        if buf[i + 4] == 2:
            out.append(i)
        i = buf.find(b"OHDR", i + 4)
    return out


def find_object(buf, name):
    """Object-header address of the hard link called `name` (link message: ... name | 8-byte address)."""
    heads = set(object_headers(buf))
    key = name.encode()
    i = buf.find(key)
    hits = []
    while i >= 0:
        addr = _u(buf, i + len(key), 8)
        # the byte before the name is its length (1-byte length field, the only size these small names need)
        if addr in heads and buf[i - 1] == len(key):
            hits.append(addr)
        i = buf.find(key, i + 1)
    if len(set(hits)) != 1:
        raise KeyError(f"{name}: {len(set(hits))} candidate links")
    return hits[0]


def _messages(buf, addr):
    """(type, data) of every message of a version-2 object header, continuation blocks included."""
    assert buf[addr:addr + 4] == b"OHDR" and buf[addr + 4] == 2
    flags = buf[addr + 5]
    p = addr + 6
    if flags & 0x20:
        p += 16                                  # access / modification / change / birth times
    if flags & 0x10:
        p += 4                                   # max compact / min dense attributes
    nsz = 1 << (flags & 3)
    size0 = _u(buf, p, nsz)
    p += nsz
    blocks = [(p, p + size0)]
    msgs = []
    while blocks:
        p, end = blocks.pop(0)
        while p + 4 <= end:
            mtype, msize = buf[p], _u(buf, p + 1, 2)
            p += 4
            if flags & 0x04:
                p += 2                           # creation order
            data = buf[p:p + msize]
            p += msize
            if mtype == 0x10:                    # continuation: offset, length of an OCHK block
                off, length = _u(data, 0, 8), _u(data, 8, 8)
                assert buf[off:off + 4] == b"OCHK"
                blocks.append((off + 4, off + length - 4))   # minus the checksum
            elif mtype != 0:
                msgs.append((mtype, data))
    return msgs


def _dtype(data):
    cls, bits0 = data[0] & 0x0F, data[1]
    size = _u(data, 4, 4)
    order = ">" if bits0 & 1 else "<"
    if cls == 0:                                 # fixed point
        return np.dtype(f"{order}{'i' if bits0 & 8 else 'u'}{size}")
    if cls == 1:
        return np.dtype(f"{order}f{size}")
    raise NotImplementedError(f"datatype class {cls}")


def _chunks(buf, addr, rank):
    """(offsets, filter_mask, address, size) of every chunk under a version-1 B-tree node."""
    assert buf[addr:addr + 4] == b"TREE" and buf[addr + 4] == 1
    level, used = buf[addr + 5], _u(buf, addr + 6, 2)
    p = addr + 24
    key = 8 + 8 * (rank + 1)
    for _ in range(used):
        size, mask = _u(buf, p, 4), _u(buf, p + 4, 4)
        offs = tuple(_u(buf, p + 8 + 8 * d, 8) for d in range(rank))
        child = _u(buf, p + key, 8)
        p += key + 8
        if level == 0:
            yield offs, mask, child, size
        else:
            yield from _chunks(buf, child, rank)


def read_dataset(buf, addr):
    """The dataset whose object header sits at `addr`, as a numpy array in its stored dtype (no scale / fill decoding)."""
    shape = dtype = layout = None
    filters = []
    for mtype, d in _messages(buf, addr):
        if mtype == 0x01:                        # dataspace
            ver, rank, fl = d[0], d[1], d[2]
            p = 4 if ver == 2 else 8
            shape = tuple(_u(d, p + 8 * k, 8) for k in range(rank))
        elif mtype == 0x03:
            dtype = _dtype(d)
        elif mtype == 0x08:                      # layout
            assert d[0] == 3, "layout message version 3 expected"
            if d[1] == 1:                        # contiguous
                layout = ("contiguous", _u(d, 2, 8), _u(d, 10, 8))
            elif d[1] == 2:
                nd = d[2]
                layout = ("chunked", _u(d, 3, 8), tuple(_u(d, 11 + 4 * k, 4) for k in range(nd - 1)))
            else:
                raise NotImplementedError("compact layout")
        elif mtype == 0x0B:                      # filter pipeline
            ver, n = d[0], d[1]
            p = 2 if ver == 2 else 8
            for _ in range(n):
                fid = _u(d, p, 2)
                p += 2
                nlen = 0
                if ver == 1 or fid >= 256:
                    nlen = _u(d, p, 2)
                    p += 2
                p += 2                           # flags
                ncv = _u(d, p, 2)
                p += 2 + nlen
                if ver == 1 and nlen % 8:
                    p += 8 - nlen % 8
                cvals = [_u(d, p + 4 * k, 4) for k in range(ncv)]
                p += 4 * ncv
                if ver == 1 and ncv % 2:
                    p += 4
                filters.append((fid, cvals))
    assert shape is not None and dtype is not None and layout is not None
    if layout[0] == "contiguous":
        _, off, size = layout
        if off == UNDEF:
            return np.zeros(shape, dtype)
        return np.frombuffer(buf, dtype, int(np.prod(shape)), off).reshape(shape).copy()
    _, btree, cdims = layout
    out = np.zeros(shape, dtype)
    for offs, mask, caddr, csize in _chunks(buf, btree, len(shape)):
        raw = bytes(buf[caddr:caddr + csize])
        for k in range(len(filters) - 1, -1, -1):           # undo the pipeline back to front
            if mask & (1 << k):
                continue
            fid, cv = filters[k]
            if fid == 1:
                raw = zlib.decompress(raw)
            elif fid == 2:                                  # shuffle: byte planes -> elements
                es = cv[0] if cv else dtype.itemsize
                n = len(raw) // es
                raw = np.frombuffer(raw, np.uint8)[: n * es].reshape(es, n).T.tobytes()
            elif fid == 3:
                raw = raw[:-4]                              # fletcher32 checksum
            else:
                raise NotImplementedError(f"filter {fid}")
        chunk = np.frombuffer(raw, dtype, int(np.prod(cdims))).reshape(cdims)
        sl = tuple(slice(o, min(o + c, s)) for o, c, s in zip(offs, cdims, shape))
        out[sl] = chunk[tuple(slice(0, s.stop - s.start) for s in sl)]
    return out


def read(path, names):
    buf = open(path, "rb").read()
    return {n: read_dataset(buf, find_object(buf, n)) for n in names}
