"""MRC / MRCS stack codec for particle stacks (reference src/mrc.py): 1024-byte little-endian header + optional
extended header + a flat array of nz images of ny x nx pixels.

Built around ONE numpy structured dtype for the header (the reference packs/unpacks with `struct`), so that a stack
can be memory-mapped and sharded by rank without copying: `open_stack(path)` returns a read-only memmap view
(n_images, ny, nx) over the payload and `read_shard(path, rank, world)` loads only this rank's contiguous slice --
the layout is a fixed header followed by images in order, so an image range is one contiguous byte range.
`parse(content)` and `write(f, array, ...)` keep the reference call signatures and header field names
(nx ny nz mode ... next ... rms nlabl labels) for drop-in use by train_particles.py:454-461.
"""
from __future__ import annotations

import numpy as np

HEADER_DTYPE = np.dtype([
    ('nx', '<i4'), ('ny', '<i4'), ('nz', '<i4'), ('mode', '<i4'),
    ('nxstart', '<i4'), ('nystart', '<i4'), ('nzstart', '<i4'),
    ('mx', '<i4'), ('my', '<i4'), ('mz', '<i4'),
    ('xlen', '<f4'), ('ylen', '<f4'), ('zlen', '<f4'),
    ('alpha', '<f4'), ('beta', '<f4'), ('gamma', '<f4'),
    ('mapc', '<i4'), ('mapr', '<i4'), ('maps', '<i4'),
    ('amin', '<f4'), ('amax', '<f4'), ('amean', '<f4'),
    ('ispg', '<i4'), ('next', '<i4'), ('creatid', '<i2'), ('_pad0', 'V30'),
    ('nint', '<i2'), ('nreal', '<i2'), ('_pad1', 'V20'),
    ('imodStamp', '<i4'), ('imodFlags', '<i4'),
    ('idtype', '<i2'), ('lens', '<i2'), ('nd1', '<i2'), ('nd2', '<i2'), ('vd1', '<i2'), ('vd2', '<i2'),
    ('tilt_ox', '<f4'), ('tilt_oy', '<f4'), ('tilt_oz', '<f4'), ('tilt_cx', '<f4'), ('tilt_cy', '<f4'), ('tilt_cz', '<f4'),
    ('xorg', '<f4'), ('yorg', '<f4'), ('zorg', '<f4'),
    ('cmap', 'S4'), ('stamp', 'S4'), ('rms', '<f4'), ('nlabl', '<i4'), ('labels', 'S800'),
])
assert HEADER_DTYPE.itemsize == 1024

MODE_DTYPES = {0: np.dtype('i1'), 1: np.dtype('<i2'), 2: np.dtype('<f4'), 3: np.dtype('2<i2'), 4: np.dtype('<c8'),
               6: np.dtype('<u2'), 16: np.dtype('3u1')}


class MRCHeader:
    """Attribute view of the 1024-byte header (same field names as the reference namedtuple)."""

    def __init__(self, rec):
        self._rec = rec

    def __getattr__(self, k):
        if k.startswith('_'):
            raise AttributeError(k)
        v = self._rec[k]
        return v.item() if hasattr(v, 'item') else v

    def tobytes(self):
        return self._rec.tobytes()


def _mode_of(dtype) -> int:
    dt = np.dtype(dtype)
    for m, d in MODE_DTYPES.items():
        if dt == d:
            return m
    raise TypeError('MRC incompatible dtype: ' + str(dtype))


def parse_header(buf) -> MRCHeader:
    return MRCHeader(np.frombuffer(buf[:1024], dtype=HEADER_DTYPE, count=1)[0])


def parse(content):
    """bytes -> (array (nz, ny, nx) [or (ny, nx) when nz == 1], header, extended_header)."""
    header = parse_header(content)
    start = 1024 + header.next
    ext = content[1024:start]
    dt = MODE_DTYPES[header.mode]
    arr = np.frombuffer(content, dtype=dt, offset=start).reshape(header.nz, header.ny, header.nx, *dt.shape)
    if header.nz == 1:
        arr = arr[0]
    return arr, header, ext


def make_header(shape, cella=(1, 1, 1), cellb=(0, 0, 0), mz=1, dtype=np.float32, dmin=0, dmax=-1, dmean=-2, rms=-1,
                exthd_size=0, ispg=0) -> MRCHeader:
    rec = np.zeros((), dtype=HEADER_DTYPE)
    rec['nx'], rec['ny'], rec['nz'] = shape[2], shape[1], shape[0]
    rec['mode'] = _mode_of(dtype)
    rec['mx'], rec['my'], rec['mz'] = 1, 1, mz
    rec['xlen'], rec['ylen'], rec['zlen'] = cella
    rec['alpha'], rec['beta'], rec['gamma'] = cellb
    rec['mapc'], rec['mapr'], rec['maps'] = 1, 2, 3
    rec['amin'], rec['amax'], rec['amean'], rec['rms'] = dmin, dmax, dmean, rms
    rec['ispg'], rec['next'] = ispg, exthd_size
    return MRCHeader(rec)


def write(f, array, header=None, extended_header=b'', ax=1, ay=1, az=1, alpha=0, beta=0, gamma=0):
    """Write a (nz, ny, nx) stack; without a header, a mode-2 (float32) header with data statistics is generated."""
    if header is None:
        array = np.ascontiguousarray(array)
        header = make_header(array.shape, (ax, ay, az), (alpha, beta, gamma), dtype=np.float32, dmin=array.min(),
                             dmax=array.max(), dmean=array.mean(), rms=array.std(), exthd_size=len(extended_header))
    f.write(header.tobytes())
    f.write(extended_header)
    f.write(np.ascontiguousarray(array).tobytes())


def open_stack(path):
    """Zero-copy read-only view (n_images, ny, nx) of an .mrc / .mrcs file + its header."""
    with open(path, 'rb') as fh:
        header = parse_header(fh.read(1024))
    dt = MODE_DTYPES[header.mode]
    mm = np.memmap(path, dtype=dt, mode='r', offset=1024 + header.next, shape=(header.nz, header.ny, header.nx) + dt.shape)
    return mm, header


def read_shard(path, rank=0, world=1):
    """This rank's contiguous slice of the stack as float32 (images are independent)."""
    mm, header = open_stack(path)
    n = mm.shape[0]
    base, extra = divmod(n, world)
    lo = rank * base + min(rank, extra)
    hi = lo + base + (1 if rank < extra else 0)
    return np.asarray(mm[lo:hi], dtype=np.float32), (lo, hi, n)
