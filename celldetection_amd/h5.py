"""HDF5 result files of the inference script: ``cd.to_h5`` / ``cd.from_h5`` (celldetection/util/util.py:1357-1400 and
:1403ff; written by celldetection_scripts/cpn_inference.py:822-823 as datasets ``contours, boxes, scores, classes,
locations, fourier, contour_proposals`` [+ labels] with the run arguments as a JSON string attribute of ``contours``).

The reference goes through h5py, which the target image does not ship; the HDF5 C library itself (libhdf5) is present,
so the same files are written through its C API with ctypes: datasets of the numpy dtype (contiguous, or chunked /
gzip-compressed through the dataset-creation property list like h5py's ``chunks=`` / ``compression=``), string attributes as
fixed-length ASCII -- readable by h5py / any HDF5 tool.  Host-side I/O only (results are copied off the GPU once per slide).
"""
import ctypes
import ctypes.util
import glob
import json
import os
from ctypes import POINTER, c_char_p, c_int, c_int64, c_size_t, c_uint, c_uint64, c_void_p

import numpy as np

__all__ = ['to_h5', 'from_h5', 'hdf5_available', 'dataset_layout', 'guess_chunk']

_H = None
_TYPES = {'float32': 'H5T_NATIVE_FLOAT_g', 'float64': 'H5T_NATIVE_DOUBLE_g', 'int8': 'H5T_NATIVE_INT8_g',
          'uint8': 'H5T_NATIVE_UINT8_g', 'int16': 'H5T_NATIVE_INT16_g', 'uint16': 'H5T_NATIVE_UINT16_g',
          'int32': 'H5T_NATIVE_INT32_g', 'uint32': 'H5T_NATIVE_UINT32_g', 'int64': 'H5T_NATIVE_INT64_g',
          'uint64': 'H5T_NATIVE_UINT64_g'}
_F_TRUNC, _F_RDONLY, _F_RDWR = 2, 0, 1
_T_INTEGER, _T_FLOAT, _T_STRING = 0, 1, 3


def _find():
    cands = [os.environ.get('CPN_HDF5_LIB'), ctypes.util.find_library('hdf5')]
    for pat in ('/opt/conda/lib/libhdf5.so*', '/usr/lib/x86_64-linux-gnu/libhdf5*.so*', '/usr/lib/x86_64-linux-gnu/hdf5/serial/libhdf5.so*',
                '/usr/local/lib/libhdf5.so*'):
        cands += sorted(glob.glob(pat))
    for c in cands:
        if not c:
            continue
        try:
            return ctypes.CDLL(c)
        except OSError:
            continue
    return None


def _lib():
    global _H
    if _H is None:
        lib = _find()
        if lib is None:
            raise RuntimeError('libhdf5 not found (set CPN_HDF5_LIB to the shared library): HDF5 export is unavailable')
        hid = c_int64
        sig = dict(H5open=(c_int, []), H5Fcreate=(hid, [c_char_p, c_uint, hid, hid]), H5Fopen=(hid, [c_char_p, c_uint, hid]),
                   H5Fclose=(c_int, [hid]), H5Screate_simple=(hid, [c_int, POINTER(c_uint64), POINTER(c_uint64)]),
                   H5Screate=(hid, [c_int]), H5Sclose=(c_int, [hid]),
                   H5Dcreate2=(hid, [hid, c_char_p, hid, hid, hid, hid, hid]), H5Dopen2=(hid, [hid, c_char_p, hid]),
                   H5Dwrite=(c_int, [hid, hid, hid, hid, hid, c_void_p]), H5Dread=(c_int, [hid, hid, hid, hid, hid, c_void_p]),
                   H5Dclose=(c_int, [hid]), H5Dget_space=(hid, [hid]), H5Dget_type=(hid, [hid]),
                   H5Sget_simple_extent_ndims=(c_int, [hid]),
                   H5Sget_simple_extent_dims=(c_int, [hid, POINTER(c_uint64), POINTER(c_uint64)]),
                   H5Tget_class=(c_int, [hid]), H5Tget_size=(c_size_t, [hid]), H5Tget_sign=(c_int, [hid]),
                   H5Tcopy=(hid, [hid]), H5Tset_size=(c_int, [hid, c_size_t]), H5Tclose=(c_int, [hid]),
                   H5Acreate2=(hid, [hid, c_char_p, hid, hid, hid, hid]), H5Awrite=(c_int, [hid, hid, c_void_p]),
                   H5Aopen=(hid, [hid, c_char_p, hid]), H5Aread=(c_int, [hid, hid, c_void_p]), H5Aget_type=(hid, [hid]),
                   H5Aclose=(c_int, [hid]), H5Aexists=(c_int, [hid, c_char_p]), H5Lexists=(c_int, [hid, c_char_p, hid]),
                   H5Ldelete=(c_int, [hid, c_char_p, hid]), H5Eset_auto2=(c_int, [hid, c_void_p, c_void_p]),
                   H5Pcreate=(hid, [hid]), H5Pclose=(c_int, [hid]), H5Pset_chunk=(c_int, [hid, c_int, POINTER(c_uint64)]),
                   H5Pset_deflate=(c_int, [hid, c_uint]), H5Zfilter_avail=(c_int, [c_int]),
                   H5Dget_create_plist=(hid, [hid]), H5Pget_layout=(c_int, [hid]), H5Pget_nfilters=(c_int, [hid]),
                   H5Pget_chunk=(c_int, [hid, c_int, POINTER(c_uint64)]), H5Dget_storage_size=(c_uint64, [hid]))
        for name, (res, args) in sig.items():
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
        if lib.H5open() < 0:
            raise RuntimeError('H5open failed')
        # hid_t is a 64-bit integer since HDF5 1.10 (a 32-bit int before): this binding declares it as int64
        maj, mnr, rel = c_uint(0), c_uint(0), c_uint(0)
        lib.H5get_libversion.argtypes = [POINTER(c_uint)] * 3
        if lib.H5get_libversion(maj, mnr, rel) < 0 or (maj.value, mnr.value) < (1, 10):
            raise RuntimeError(f'libhdf5 {maj.value}.{mnr.value}.{rel.value} is too old: HDF5 >= 1.10 (64-bit hid_t) is required')
        lib.H5Eset_auto2(0, None, None)  # no error stack printing: failures are raised as Python exceptions
        _H = lib
    return _H


def hdf5_available() -> bool:
    try:
        _lib()
        return True
    except RuntimeError:
        return False


def _tid(lib, dtype):
    name = _TYPES.get(np.dtype(dtype).name)
    if name is None:
        raise TypeError(f'unsupported dtype for HDF5 export: {dtype}')
    return c_int64.in_dll(lib, name).value


def _check(v, what):
    if v < 0:
        raise OSError(f'HDF5: {what} failed')
    return v


def _np(v):
    if hasattr(v, 'detach'):
        v = v.detach().cpu().numpy()
    v = np.asarray(v)
    if v.dtype == np.bool_:
        v = v.astype(np.uint8)
    return np.ascontiguousarray(v)


_CHUNK_BASE, _CHUNK_MIN, _CHUNK_MAX = 16 * 1024, 8 * 1024, 1024 * 1024


def guess_chunk(shape, typesize):
    """h5py's auto-chunking (``chunks=True``, or a filter without explicit chunks; h5py/_hl/filters.py ``guess_chunk``,
    restated -- h5py is absent here, so the chunk SHAPE is unpinned; it does not change what a reader gets back): halve the
    dimensions in turn until a chunk holds about 16 KiB * 2^log10(dataset MiB), clamped to [8 KiB, 1 MiB]."""
    chunks = np.array([x if x != 0 else 1024 for x in shape], dtype='=f8')
    ndims = len(chunks)
    if ndims == 0:
        raise ValueError('Chunks not allowed for scalar datasets.')
    dset_size = np.prod(chunks) * typesize
    target = _CHUNK_BASE * (2 ** np.log10(dset_size / (1024. * 1024)))
    target = min(max(target, _CHUNK_MIN), _CHUNK_MAX)
    idx = 0
    while True:
        chunk_bytes = np.prod(chunks) * typesize
        if (chunk_bytes < target or abs(chunk_bytes - target) / target < .5) and chunk_bytes < _CHUNK_MAX:
            break
        if np.prod(chunks) == 1:
            break
        chunks[idx % ndims] = np.ceil(chunks[idx % ndims] / 2.)
        idx += 1
    return tuple(int(x) for x in chunks)


def _creation_plist(lib, arr, key, chunks, compression):
    """Dataset-creation property list for h5py's ``chunks`` / ``compression`` arguments (0 = default: contiguous)."""
    chunks_ = chunks[key] if isinstance(chunks, dict) else chunks
    if isinstance(chunks_, (int, np.integer)) and not isinstance(chunks_, bool) and arr.ndim > 1:
        chunks_ = tuple(int(v) for v in np.minimum((256,) * arr.ndim, arr.shape))  # util.py:1387-1388, as written there
    level = None
    if compression is not None and compression is not False:
        if compression == 'gzip' or compression is True:
            level = 4  # h5py's default gzip level
        elif isinstance(compression, (int, np.integer)) and 0 <= int(compression) < 10:
            level = int(compression)
        else:
            raise NotImplementedError(f'to_h5: compression {compression!r} needs a filter libhdf5 does not ship '
                                      "(supported: 'gzip' or a gzip level 0..9)")
        if lib.H5Zfilter_avail(1) <= 0:  # H5Z_FILTER_DEFLATE
            raise RuntimeError('to_h5: this libhdf5 was built without the deflate (gzip) filter')
    if chunks_ is None and level is None:
        return 0
    if arr.ndim == 0:
        raise TypeError("Scalar datasets don't support chunk/filter options")
    if arr.size == 0:  # (a chunk dimension must be positive; nothing to lay out)
        return 0
    if chunks_ is None or chunks_ is True:
        chunks_ = guess_chunk(arr.shape, arr.dtype.itemsize)
    if isinstance(chunks_, (int, np.integer)):
        chunks_ = (int(chunks_),)
    chunks_ = tuple(int(c) for c in chunks_)
    if len(chunks_) != arr.ndim or any(c < 1 for c in chunks_):
        raise ValueError(f'to_h5: chunk shape {chunks_} does not fit dataset {key!r} of shape {arr.shape}')
    if any(c > d for c, d in zip(chunks_, arr.shape)):
        raise ValueError(f'to_h5: chunk shape {chunks_} must not exceed the (fixed) dataset shape {arr.shape}')
    pl = _check(lib.H5Pcreate(c_int64.in_dll(lib, 'H5P_CLS_DATASET_CREATE_ID_g').value), 'property list')
    _check(lib.H5Pset_chunk(pl, arr.ndim, (c_uint64 * arr.ndim)(*chunks_)), 'set chunk shape')
    if level is not None:
        _check(lib.H5Pset_deflate(pl, level), 'set gzip level')
    return pl


def dataset_layout(filename, key):
    """-> dict(chunks = chunk shape or None, filters = number of filters, storage_bytes) of a dataset (tests / inspection)."""
    lib = _lib()
    f = _check(lib.H5Fopen(os.fsencode(filename), _F_RDONLY, 0), f'open {filename}')
    try:
        ds = _check(lib.H5Dopen2(f, key.encode(), 0), f'open dataset {key}')
        pl, space = lib.H5Dget_create_plist(ds), lib.H5Dget_space(ds)
        nd = lib.H5Sget_simple_extent_ndims(space)
        chunks = None
        if lib.H5Pget_layout(pl) == 2:  # H5D_CHUNKED
            dims = (c_uint64 * max(nd, 1))()
            lib.H5Pget_chunk(pl, max(nd, 1), dims)
            chunks = tuple(int(d) for d in dims[:nd])
        out = dict(chunks=chunks, filters=int(lib.H5Pget_nfilters(pl)), storage_bytes=int(lib.H5Dget_storage_size(ds)))
        lib.H5Sclose(space)
        lib.H5Pclose(pl)
        lib.H5Dclose(ds)
        return out
    finally:
        lib.H5Fclose(f)


def to_h5(filename, mode='w', chunks=None, compression=None, overwrite=False, driver=None, create_dataset_kw=None,
          attributes=None, **kwargs):
    """``cd.to_h5``: writes ``{dataset_name: array}`` (numpy arrays or tensors) and ``attributes``
    (``{dataset_name: {attribute: value}}``; str / numbers / dicts as JSON text) to an HDF5 file.  ``chunks`` (shape, True =
    auto, int, or a dict per dataset) and ``compression`` ('gzip' or a gzip level 0..9) as in the reference
    (util/util.py:1385-1395 -> ``h5py create_dataset``)."""
    lib = _lib()
    attributes = attributes or {}
    fn = os.fsencode(filename)
    # h5py.File mode semantics: 'w' create / truncate; 'w-' and 'x' create, fail if the file exists; 'r+' read/write, the
    # file must exist; 'a' read/write if it exists, create otherwise
    exists = os.path.isfile(filename)
    if mode not in ('w', 'w-', 'x', 'r+', 'a'):
        raise ValueError(f"to_h5: mode must be one of 'w', 'w-', 'x', 'r+', 'a' (got {mode!r})")
    if mode in ('w-', 'x') and exists:
        raise FileExistsError(f'to_h5: {filename} exists (mode {mode!r})')
    if mode == 'r+' and not exists:
        raise FileNotFoundError(f'to_h5: {filename} does not exist (mode {mode!r})')
    if driver is not None or create_dataset_kw:
        import warnings
        warnings.warn('to_h5: driver / create_dataset_kw are not supported by the libhdf5 binding and ignored',
                      RuntimeWarning, stacklevel=2)
    if mode in ('w', 'w-', 'x') or not exists:
        f = _check(lib.H5Fcreate(fn, _F_TRUNC, 0, 0), f'create {filename}')
    else:
        f = _check(lib.H5Fopen(fn, _F_RDWR, 0), f'open {filename}')
    try:
        for key, value in kwargs.items():
            if value is None:
                continue
            arr = _np(value)
            k = key.encode()
            if lib.H5Lexists(f, k, 0) > 0:  # the reference replaces the contents; shapes may differ here -> recreate
                _check(lib.H5Ldelete(f, k, 0), f'delete {key}')
            dims = (c_uint64 * max(arr.ndim, 1))(*arr.shape)
            space = _check(lib.H5Screate_simple(arr.ndim, dims, None) if arr.ndim else lib.H5Screate(0), 'dataspace')
            tid = _tid(lib, arr.dtype)
            dcpl = _creation_plist(lib, arr, key, chunks, compression)
            try:
                ds = _check(lib.H5Dcreate2(f, k, tid, space, 0, dcpl, 0), f'create dataset {key}')
            finally:
                if dcpl:
                    lib.H5Pclose(dcpl)
            try:
                if arr.size:
                    _check(lib.H5Dwrite(ds, tid, 0, 0, 0, arr.ctypes.data_as(c_void_p)), f'write {key}')
                for an, av in (attributes.get(key) or {}).items():
                    _write_attr(lib, ds, an, av)
            finally:
                lib.H5Dclose(ds)
                lib.H5Sclose(space)
    finally:
        lib.H5Fclose(f)
    return filename


def _write_attr(lib, obj, name, value):
    if isinstance(value, (dict, list, tuple)):
        value = json.dumps(value)
    space = _check(lib.H5Screate(0), 'scalar dataspace')
    try:
        if isinstance(value, (str, bytes)):
            raw = value.encode() if isinstance(value, str) else value
            t = _check(lib.H5Tcopy(c_int64.in_dll(lib, 'H5T_C_S1_g').value), 'string type')
            lib.H5Tset_size(t, max(len(raw), 1))
            a = _check(lib.H5Acreate2(obj, name.encode(), t, space, 0, 0), f'attribute {name}')
            buf = ctypes.create_string_buffer(raw, max(len(raw), 1))
            _check(lib.H5Awrite(a, t, buf), f'write attribute {name}')
            lib.H5Aclose(a)
            lib.H5Tclose(t)
        else:
            arr = np.asarray(value)
            arr = arr.astype(np.float64 if arr.dtype.kind == 'f' else np.int64).reshape(())
            tid = _tid(lib, arr.dtype)
            a = _check(lib.H5Acreate2(obj, name.encode(), tid, space, 0, 0), f'attribute {name}')
            _check(lib.H5Awrite(a, tid, arr.ctypes.data_as(c_void_p)), f'write attribute {name}')
            lib.H5Aclose(a)
    finally:
        lib.H5Sclose(space)


def _dtype_of(lib, t):
    cls, size = lib.H5Tget_class(t), int(lib.H5Tget_size(t))
    if cls == _T_FLOAT:
        return np.dtype(f'f{size}')
    if cls == _T_INTEGER:
        return np.dtype(('i' if lib.H5Tget_sign(t) == 1 else 'u') + str(size))
    raise TypeError('only integer / float datasets are supported')


def from_h5(filename, *keys, attributes=False):
    """Reads datasets (all of them given no ``keys`` is not supported: name them) -> single array or tuple of arrays;
    ``attributes=True`` returns ``(arrays..., {key: {attr: str}})`` for the string attributes written by ``to_h5``."""
    lib = _lib()
    f = _check(lib.H5Fopen(os.fsencode(filename), _F_RDONLY, 0), f'open {filename}')
    out, attrs = [], {}
    try:
        for key in keys:
            ds = _check(lib.H5Dopen2(f, key.encode(), 0), f'open dataset {key}')
            try:
                space, t = lib.H5Dget_space(ds), lib.H5Dget_type(ds)
                nd = lib.H5Sget_simple_extent_ndims(space)
                dims = (c_uint64 * max(nd, 1))()
                if nd:
                    lib.H5Sget_simple_extent_dims(space, dims, None)
                arr = np.empty(tuple(int(d) for d in dims[:nd]), _dtype_of(lib, t))
                if arr.size:
                    _check(lib.H5Dread(ds, _tid(lib, arr.dtype), 0, 0, 0, arr.ctypes.data_as(c_void_p)), f'read {key}')
                out.append(arr)
                lib.H5Tclose(t)
                lib.H5Sclose(space)
                if attributes and lib.H5Aexists(ds, b'args') > 0:
                    a = lib.H5Aopen(ds, b'args', 0)
                    at = lib.H5Aget_type(a)
                    buf = ctypes.create_string_buffer(int(lib.H5Tget_size(at)))
                    lib.H5Aread(a, at, buf)
                    attrs.setdefault(key, {})['args'] = buf.raw.rstrip(b'\0').decode()
                    lib.H5Tclose(at)
                    lib.H5Aclose(a)
            finally:
                lib.H5Dclose(ds)
    finally:
        lib.H5Fclose(f)
    res = out[0] if len(out) == 1 and not attributes else tuple(out)
    return (res + (attrs,)) if attributes else res
