"""HDF5 result files of the inference script: ``cd.to_h5`` / ``cd.from_h5`` (celldetection/util/util.py:1357-1400 and
:1403ff; written by celldetection_scripts/cpn_inference.py:822-823 as datasets ``contours, boxes, scores, classes,
locations, fourier, contour_proposals`` [+ labels] with the run arguments as a JSON string attribute of ``contours``).

The reference goes through h5py, which the target image does not ship; the HDF5 C library itself (libhdf5) is present,
so the same files are written through its C API with ctypes: datasets of the numpy dtype (contiguous, or chunked /
gzip-compressed through the dataset-creation property list like h5py's ``chunks=`` / ``compression=``).  Attributes follow
h5py's documented type mapping in both directions (``str`` <-> variable-length UTF-8 string on a scalar dataspace, ``bytes``
-> variable-length ASCII, ``numpy.bytes_`` <-> fixed-length string, Python / numpy numbers and arrays <-> native numeric
types, ``bool`` <-> the int8 enum ``{FALSE, TRUE}``), so the ``args`` attribute the reference script writes
(cpn_inference.py:822-823: a JSON ``str``) reads back as ``str`` here and a file written here reads back as ``str`` in
h5py.  Host-side I/O only (results are copied off the GPU once per slide).
"""
import ctypes
import ctypes.util
import glob
import json
import os
from ctypes import POINTER, c_char_p, c_int, c_int64, c_size_t, c_ssize_t, c_uint, c_uint64, c_void_p

import numpy as np

__all__ = ['to_h5', 'from_h5', 'hdf5_available', 'dataset_layout', 'guess_chunk']

_H = None
_TYPES = {'float32': 'H5T_NATIVE_FLOAT_g', 'float64': 'H5T_NATIVE_DOUBLE_g', 'int8': 'H5T_NATIVE_INT8_g',
          'uint8': 'H5T_NATIVE_UINT8_g', 'int16': 'H5T_NATIVE_INT16_g', 'uint16': 'H5T_NATIVE_UINT16_g',
          'int32': 'H5T_NATIVE_INT32_g', 'uint32': 'H5T_NATIVE_UINT32_g', 'int64': 'H5T_NATIVE_INT64_g',
          'uint64': 'H5T_NATIVE_UINT64_g'}
_F_TRUNC, _F_RDONLY, _F_RDWR = 2, 0, 1
_T_INTEGER, _T_FLOAT, _T_STRING, _T_ENUM, _T_ARRAY = 0, 1, 3, 8, 10
_CSET_ASCII, _CSET_UTF8 = 0, 1
_VARIABLE = c_size_t(-1).value  # H5T_VARIABLE


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
                   H5Pget_chunk=(c_int, [hid, c_int, POINTER(c_uint64)]), H5Dget_storage_size=(c_uint64, [hid]),
                   H5Tis_variable_str=(c_int, [hid]), H5Tset_cset=(c_int, [hid, c_int]), H5Tget_cset=(c_int, [hid]),
                   H5Tset_strpad=(c_int, [hid, c_int]), H5Tget_super=(hid, [hid]),
                   H5Tenum_create=(hid, [hid]), H5Tenum_insert=(c_int, [hid, c_char_p, c_void_p]),
                   H5Tget_nmembers=(c_int, [hid]), H5Aget_space=(hid, [hid]),
                   H5Aget_name=(c_ssize_t, [hid, c_size_t, c_char_p]),
                   H5Aopen_by_idx=(hid, [hid, c_char_p, c_int, c_int, c_uint64, hid, hid]),
                   H5Adelete=(c_int, [hid, c_char_p]),
                   H5Aiterate2=(c_int, [hid, c_int, c_int, POINTER(c_uint64), c_void_p, c_void_p]),
                   H5Gget_info=(c_int, [hid, c_void_p]),
                   H5Sselect_hyperslab=(c_int, [hid, c_int, POINTER(c_uint64), POINTER(c_uint64), POINTER(c_uint64),
                                                POINTER(c_uint64)]),
                   H5Lget_name_by_idx=(c_ssize_t, [hid, c_char_p, c_int, c_int, c_uint64, c_char_p, c_size_t, hid]))
        for name, (res, args) in sig.items():
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
        # releases the buffers libhdf5 allocated for variable-length data: H5Treclaim since 1.12, H5Dvlen_reclaim before
        rec = getattr(lib, 'H5Treclaim', None) or lib.H5Dvlen_reclaim
        rec.restype, rec.argtypes = c_int, [hid, hid, hid, c_void_p]
        lib._cpn_reclaim = rec
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
    return np.ascontiguousarray(v).reshape(v.shape)  # (ascontiguousarray alone turns a 0-d array into a 1-d one)


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
    if isinstance(chunks_, (int, np.integer)) and arr.ndim > 1:
        # util.py:1387-1388 as written there: ``isinstance(True, int)`` holds in Python, so ``chunks=True`` on an N-D array is
        # min(256, extent) per axis as well (h5py's auto-chunking only ever sees 1-D arrays from the reference)
        chunks_ = tuple(int(v) for v in np.minimum((256,) * arr.ndim, arr.shape))
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
        pl, space = _check(lib.H5Dget_create_plist(ds), 'creation property list'), _check(lib.H5Dget_space(ds), 'dataspace')
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


def _same_shape(lib, f, k, shape):
    ds = lib.H5Dopen2(f, k, 0)
    if ds < 0:
        return False
    try:
        space = _check(lib.H5Dget_space(ds), 'dataspace')
        try:
            return _space_shape(lib, space) == tuple(shape)
        finally:
            lib.H5Sclose(space)
    finally:
        lib.H5Dclose(ds)


def to_h5(filename, mode='w', chunks=None, compression=None, overwrite=False, driver=None, create_dataset_kw=None,
          attributes=None, **kwargs):
    """``cd.to_h5``: writes ``{dataset_name: array}`` (numpy arrays or tensors) and ``attributes``
    (``{dataset_name: {attribute: value}}``, typed like ``h5py``'s ``attrs.update``: ``_write_attr``) to an HDF5 file.  ``chunks`` (shape, True =
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
            if lib.H5Lexists(f, k, 0) > 0:
                # util.py:1389-1394: an existing dataset keeps its layout and dtype and takes the new contents (``ds[:] = v``);
                # here also with ``overwrite`` unset and another shape (h5py would refuse to broadcast): it is recreated
                if not overwrite and _same_shape(lib, f, k, arr.shape):
                    ds = _check(lib.H5Dopen2(f, k, 0), f'open dataset {key}')
                    try:
                        if arr.size:
                            _check(lib.H5Dwrite(ds, _tid(lib, arr.dtype), 0, 0, 0, arr.ctypes.data_as(c_void_p)), f'write {key}')
                        for an, av in (attributes.get(key) or {}).items():
                            _write_attr(lib, ds, an, av)
                    finally:
                        lib.H5Dclose(ds)
                    continue
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


def _str_type(lib, size, cset):
    t = _check(lib.H5Tcopy(c_int64.in_dll(lib, 'H5T_C_S1_g').value), 'string type')
    _check(lib.H5Tset_size(t, size), 'string size')
    _check(lib.H5Tset_cset(t, cset), 'string character set')
    return t


def _bool_type(lib):
    """h5py's mapping of ``numpy.bool_``: an enum over int8 with the members FALSE = 0, TRUE = 1."""
    t = _check(lib.H5Tenum_create(c_int64.in_dll(lib, 'H5T_NATIVE_INT8_g').value), 'enum type')
    for name, v in ((b'FALSE', 0), (b'TRUE', 1)):
        _check(lib.H5Tenum_insert(t, name, ctypes.byref(ctypes.c_int8(v))), 'enum member')
    return t


def _write_attr(lib, obj, name, value):
    """One attribute, typed the way ``h5py``'s ``AttributeManager.__setitem__`` types it (module docstring); an existing
    attribute of that name is replaced (``attrs.update``, util/util.py:1396-1399).  dicts are stored as their JSON text
    (h5py has no mapping for them; the reference script serialises its ``args`` itself, cpn_inference.py:822-823)."""
    if value is None:
        raise TypeError(f'attribute {name!r}: None has no HDF5 type (the reference script stores its arguments as JSON for '
                        'this reason, cpn_inference.py:822)')
    if isinstance(value, dict):
        value = json.dumps(value)
    if hasattr(value, 'detach'):
        value = value.detach().cpu().numpy()
    nm = name.encode()
    if lib.H5Aexists(obj, nm) > 0:
        _check(lib.H5Adelete(obj, nm), f'replace attribute {name}')
    keep = []  # buffers the write reads from
    if isinstance(value, (str, bytes)) and not isinstance(value, np.bytes_):
        raw = value.encode('utf-8') if isinstance(value, str) else bytes(value)
        if b'\0' in raw:
            raise ValueError('variable-length strings cannot hold NUL bytes')
        t, close_t = _str_type(lib, _VARIABLE, _CSET_UTF8 if isinstance(value, str) else _CSET_ASCII), True
        shape = ()
        keep.append(ctypes.create_string_buffer(raw))
        buf = (c_char_p * 1)(ctypes.cast(keep[0], c_char_p))
    else:
        arr = np.asarray(value)
        shape = arr.shape  # (before ascontiguousarray, which turns a 0-d array into a 1-d one)
        if arr.dtype.kind == 'U' or (arr.dtype.kind == 'O' and all(isinstance(v, str) for v in arr.ravel())):
            strs = [str(v).encode('utf-8') for v in arr.ravel()]  # array of str -> array of variable-length UTF-8
            t, close_t = _str_type(lib, _VARIABLE, _CSET_UTF8), True
            keep += [ctypes.create_string_buffer(r) for r in strs]
            buf = (c_char_p * max(len(strs), 1))(*[ctypes.cast(k, c_char_p) for k in keep])
        elif arr.dtype.kind == 'S':
            arr = np.ascontiguousarray(arr)
            t, close_t = _str_type(lib, max(arr.dtype.itemsize, 1), _CSET_ASCII), True
            _check(lib.H5Tset_strpad(t, 1), 'string padding')  # H5T_STR_NULLPAD, like numpy's S dtype
            keep.append(arr)
            buf = arr.ctypes.data_as(c_void_p)
        elif arr.dtype.kind == 'b':
            arr = np.ascontiguousarray(arr.astype(np.int8))
            t, close_t = _bool_type(lib), True
            keep.append(arr)
            buf = arr.ctypes.data_as(c_void_p)
        elif arr.dtype.kind in 'iuf' and arr.dtype.name in _TYPES:
            arr = np.ascontiguousarray(arr)
            t, close_t = _tid(lib, arr.dtype), False
            keep.append(arr)
            buf = arr.ctypes.data_as(c_void_p)
        else:
            raise TypeError(f'attribute {name!r}: no HDF5 type for {type(value).__name__} of dtype {arr.dtype}')
    if len(shape):
        space = _check(lib.H5Screate_simple(len(shape), (c_uint64 * len(shape))(*shape), None), 'attribute dataspace')
    else:
        space = _check(lib.H5Screate(0), 'scalar dataspace')
    try:
        a = _check(lib.H5Acreate2(obj, nm, t, space, 0, 0), f'attribute {name}')
        try:
            if int(np.prod(shape, dtype=np.int64)) > 0:
                _check(lib.H5Awrite(a, t, buf), f'write attribute {name}')
        finally:
            lib.H5Aclose(a)
    finally:
        lib.H5Sclose(space)
        if close_t:
            lib.H5Tclose(t)


def _dtype_of(lib, t):
    cls, size = lib.H5Tget_class(t), int(lib.H5Tget_size(t))
    if cls == _T_FLOAT:
        return np.dtype(f'f{size}')
    if cls == _T_INTEGER:
        return np.dtype(('i' if lib.H5Tget_sign(t) == 1 else 'u') + str(size))
    raise TypeError('only integer / float datasets are supported')


def _space_shape(lib, space):
    nd = lib.H5Sget_simple_extent_ndims(space)
    if nd <= 0:
        return ()
    dims = (c_uint64 * nd)()
    lib.H5Sget_simple_extent_dims(space, dims, None)
    return tuple(int(d) for d in dims)


def _read_attr(lib, a, name):
    """Value of an open attribute with h5py's read-side mapping: variable-length strings -> ``str`` (UTF-8 decoded, either
    character set), fixed-length strings -> ``numpy.bytes_``, the int8 ``{FALSE, TRUE}`` enum -> ``numpy.bool_``, numbers ->
    numpy scalars / arrays."""
    t, space = _check(lib.H5Aget_type(a), 'attribute type'), _check(lib.H5Aget_space(a), 'attribute dataspace')
    try:
        shape = _space_shape(lib, space)
        n = int(np.prod(shape, dtype=np.int64))
        cls = lib.H5Tget_class(t)
        if cls == _T_STRING and lib.H5Tis_variable_str(t) > 0:
            mt = _str_type(lib, _VARIABLE, lib.H5Tget_cset(t))
            try:
                ptrs = (c_void_p * max(n, 1))()
                if n:
                    _check(lib.H5Aread(a, mt, ptrs), f'read attribute {name}')
                vals = [(ctypes.string_at(p) if p else b'').decode('utf-8', 'surrogateescape') for p in ptrs[:n]]
                if n:
                    lib._cpn_reclaim(mt, space, 0, ptrs)  # the char* buffers belong to libhdf5
            finally:
                lib.H5Tclose(mt)
            if shape == ():
                return vals[0]
            out = np.empty(n, dtype=object)
            out[:] = vals
            return out.reshape(shape)
        if cls == _T_STRING:
            size = int(lib.H5Tget_size(t))
            arr = np.zeros(shape, dtype=f'S{size}')
            if n:
                _check(lib.H5Aread(a, t, arr.ctypes.data_as(c_void_p)), f'read attribute {name}')
            return arr[()] if shape == () else arr
        if cls == _T_ENUM:
            base = _check(lib.H5Tget_super(t), 'enum base type')
            try:
                dt = _dtype_of(lib, base)
            finally:
                lib.H5Tclose(base)
            arr = np.zeros(shape, dtype=dt)
            if n:
                _check(lib.H5Aread(a, t, arr.ctypes.data_as(c_void_p)), f'read attribute {name}')
            if dt.itemsize == 1 and lib.H5Tget_nmembers(t) == 2:
                arr = arr.astype(np.bool_)
            return arr[()] if shape == () else arr
        if cls in (_T_INTEGER, _T_FLOAT):
            dt = _dtype_of(lib, t)
            arr = np.zeros(shape, dtype=dt)
            if n:
                _check(lib.H5Aread(a, _tid(lib, dt), arr.ctypes.data_as(c_void_p)), f'read attribute {name}')
            return arr[()] if shape == () else arr
        raise TypeError(f'attribute {name!r}: HDF5 type class {cls} is not supported')
    finally:
        lib.H5Sclose(space)
        lib.H5Tclose(t)


_ATTR_OP = ctypes.CFUNCTYPE(c_int, c_int64, c_char_p, c_void_p, c_void_p)


def _read_attrs(lib, obj):
    """``dict(obj.attrs)``: every attribute of an open object, in name order."""
    names = []

    def op(_loc, name, _info, _data):
        names.append(bytes(name))
        return 0

    cb = _ATTR_OP(op)
    _check(lib.H5Aiterate2(obj, 0, 0, None, ctypes.cast(cb, c_void_p), None), 'iterate attributes')  # H5_INDEX_NAME, INC
    out = {}
    for nm in names:
        a = _check(lib.H5Aopen(obj, nm, 0), f'open attribute {nm!r}')
        try:
            out[nm.decode('utf-8')] = _read_attr(lib, a, nm.decode('utf-8'))
        finally:
            lib.H5Aclose(a)
    return out


class _GInfo(ctypes.Structure):  # H5G_info_t
    _fields_ = [('storage_type', c_int), ('nlinks', c_uint64), ('max_corder', c_int64), ('mounted', c_uint)]


def _keys(lib, f):
    info = _GInfo()
    _check(lib.H5Gget_info(f, ctypes.byref(info)), 'group info')
    out = []
    for i in range(int(info.nlinks)):
        n = _check(lib.H5Lget_name_by_idx(f, b'.', 0, 0, i, None, 0, 0), 'link name')
        buf = ctypes.create_string_buffer(n + 1)
        lib.H5Lget_name_by_idx(f, b'.', 0, 0, i, buf, n + 1, 0)
        out.append(buf.value.decode('utf-8'))
    return out


def _plan_index(index, shape):
    """numpy-style basic index -> (start, stride, count per dataset axis, result shape), or None when the index needs
    more than one hyperslab (lists, masks, negative steps): those read the dataset and index it in memory."""
    if not isinstance(index, tuple):
        index = (index,)
    if any(i is Ellipsis for i in index):
        e = [k for k, i in enumerate(index) if i is Ellipsis]
        if len(e) > 1:
            raise IndexError("an index can only have a single ellipsis ('...')")
        fill = len(shape) - (len(index) - 1)
        if fill < 0:
            raise IndexError('too many indices for the dataset')
        index = index[:e[0]] + (slice(None),) * fill + index[e[0] + 1:]
    if len(index) > len(shape):
        raise IndexError('too many indices for the dataset')
    index = index + (slice(None),) * (len(shape) - len(index))
    start, stride, count, out_shape = [], [], [], []
    for i, d in zip(index, shape):
        if isinstance(i, (int, np.integer)) and not isinstance(i, (bool, np.bool_)):
            j = int(i) + (d if i < 0 else 0)
            if not 0 <= j < d:
                raise IndexError(f'index {int(i)} out of range for an axis of length {d}')
            start.append(j), stride.append(1), count.append(1)
        elif isinstance(i, slice):
            b, e, st = i.indices(d)
            if st < 1:
                return None
            c = max(0, -(-(e - b) // st))
            start.append(b), stride.append(st), count.append(c), out_shape.append(c)
        else:
            return None
    return start, stride, count, tuple(out_shape)


def _read_dataset(lib, f, key, index=None):
    ds = _check(lib.H5Dopen2(f, key.encode(), 0), f'open dataset {key}')
    try:
        space, t = _check(lib.H5Dget_space(ds), 'dataspace'), _check(lib.H5Dget_type(ds), 'datatype')
        try:
            shape = _space_shape(lib, space)
            dt = _dtype_of(lib, t)
            if index is not None and shape == ():
                raise ValueError('Illegal slicing argument for scalar dataspace')  # (h5py's message)
            plan = None if index is None else _plan_index(index, shape)
            if plan is None:
                arr = np.empty(shape, dt)
                if arr.size:
                    _check(lib.H5Dread(ds, _tid(lib, dt), 0, 0, 0, arr.ctypes.data_as(c_void_p)), f'read {key}')
                if index is not None:
                    arr = arr[index]
            else:
                start, stride, count, out_shape = plan
                nd = len(shape)
                arr = np.empty(out_shape, dt)
                if arr.size:
                    u = c_uint64 * nd
                    _check(lib.H5Sselect_hyperslab(space, 0, u(*start), u(*stride), u(*count), None), 'select hyperslab')
                    mem = _check(lib.H5Screate_simple(nd, u(*count), None), 'memory dataspace')
                    try:
                        _check(lib.H5Dread(ds, _tid(lib, dt), mem, space, 0, arr.ctypes.data_as(c_void_p)), f'read {key}')
                    finally:
                        lib.H5Sclose(mem)
            attrs = _read_attrs(lib, ds)
        finally:
            lib.H5Tclose(t)
            lib.H5Sclose(space)
    finally:
        lib.H5Dclose(ds)
    return arr, attrs


def from_h5(filename, *keys, file_kwargs=None, driver=None, return_attrs=False, **keys_slices):
    """``cd.from_h5`` (util/util.py:1459-1488), same signature and return value: the datasets named in ``keys`` in full and
    those of ``keys_slices`` indexed (``from_h5('file.h5', 'key0', key=slice(0, 42))``: ints, slices and ``...`` become ONE
    hyperslab read, anything else is indexed in memory) -> a single array, or a tuple when more than one was asked for; with
    ``return_attrs=True`` -> ``(res, attrs)``, ``attrs`` = one ``dict(dataset.attrs)`` per returned array.  Without any key
    the available keys are printed and ``()`` is returned, as there.  ``file_kwargs`` / ``driver`` configure h5py's file
    object in the reference; the libhdf5 binding opens with the default (sec2) driver and warns about anything else.
    (``attributes`` -- this package's round-2..4 spelling with another return layout -- is no keyword any more: as in the
    reference, it would name a dataset.)"""
    lib = _lib()
    file_kwargs = dict(file_kwargs or {})
    if driver is not None:
        file_kwargs['driver'] = driver
    if any(v is not None for k, v in file_kwargs.items() if not (k == 'driver' and v in ('sec2', None))):
        import warnings
        warnings.warn(f'from_h5: file_kwargs {file_kwargs} are not supported by the libhdf5 binding and ignored',
                      RuntimeWarning, stacklevel=2)
    if not os.path.isfile(filename):
        raise FileNotFoundError(f'from_h5: {filename} does not exist')
    f = _check(lib.H5Fopen(os.fsencode(filename), _F_RDONLY, 0), f'open {filename}')
    res, attrs = [], []
    try:
        if len(keys) == 0 and len(keys_slices) == 0:
            print('Available keys:', _keys(lib, f), flush=True)
        for key, index in [(k, None) for k in keys] + list(keys_slices.items()):
            if lib.H5Lexists(f, key.encode(), 0) <= 0:
                raise KeyError(f"Unable to open object (object '{key}' doesn't exist)")
            a, at = _read_dataset(lib, f, key, index)
            res.append(a), attrs.append(at)
    finally:
        lib.H5Fclose(f)
    res = res[0] if len(res) == 1 else tuple(res)
    if return_attrs:
        return res, tuple(attrs)
    return res
