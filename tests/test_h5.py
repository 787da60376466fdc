"""HDF5 result files: to_h5 / from_h5 round trip through libhdf5's C API."""
import json

import numpy as np
import pytest
import torch

from celldetection_amd import h5


@pytest.mark.skipif(not h5.hdf5_available(), reason='libhdf5 not present')
def test_result_file_roundtrip(tmp_path):
    rng = np.random.default_rng(0)
    k, s, o = 23, 32, 5
    out = dict(contours=rng.random((k, s, 2)).astype(np.float32), boxes=rng.random((k, 4)).astype(np.float32),
               scores=torch.rand(k), classes=np.ones(k, np.int64), locations=rng.random((k, 2)).astype(np.float32),
               fourier=rng.standard_normal((k, o, 4)).astype(np.float32),
               contour_proposals=rng.random((k, s, 2)).astype(np.float32), labels=rng.integers(0, 9, (40, 50, 2)).astype(np.int32))
    args = dict(model='ginoro', tile_size=512, stride=384, nms_thresh=None)
    f = str(tmp_path / 'slide.h5')
    h5.to_h5(f, **out, attributes=dict(contours=dict(args=json.dumps(args))))  # cpn_inference.py:822-823
    assert open(f, 'rb').read(8) == b'\x89HDF\r\n\x1a\n'
    names = list(out)
    arrays, attrs = h5.from_h5(f, *names, return_attrs=True)  # util/util.py:1475-1488: (res, tuple of attr dicts)
    assert isinstance(attrs, tuple) and len(attrs) == len(names)
    attrs = dict(zip(names, attrs))
    for name, a in zip(names, arrays):
        exp = out[name].numpy() if isinstance(out[name], torch.Tensor) else out[name]
        assert a.dtype == exp.dtype and a.shape == exp.shape, name
        np.testing.assert_array_equal(a, exp)
    assert isinstance(attrs['contours']['args'], str) and json.loads(attrs['contours']['args']) == args
    assert attrs['boxes'] == {}
    # empty result sets and replacing a dataset in an existing file
    h5.to_h5(f, mode='a', scores=np.zeros((0,), np.float32), contours=np.zeros((0, s, 2), np.float32))
    sc, con, boxes = h5.from_h5(f, 'scores', 'contours', 'boxes')
    assert sc.shape == (0,) and con.shape == (0, s, 2) and boxes.shape == (k, 4)


@pytest.mark.skipif(not h5.hdf5_available(), reason='libhdf5 not present')
def test_file_modes_follow_h5py(tmp_path):
    """'w-' / 'x' must not truncate an existing file, 'r+' needs an existing one, unsupported dataset options
    are announced instead of being ignored silently."""
    f = str(tmp_path / 'm.h5')
    with pytest.raises(FileNotFoundError):
        h5.to_h5(f, mode='r+', a=np.arange(3))
    h5.to_h5(f, mode='x', a=np.arange(3))
    for mode in ('x', 'w-'):
        with pytest.raises(FileExistsError):
            h5.to_h5(f, mode=mode, a=np.arange(5))
    assert h5.from_h5(f, 'a').shape == (3,)  # untouched
    h5.to_h5(f, mode='r+', b=np.ones((2, 2), np.float32))
    h5.to_h5(f, mode='a', c=np.zeros(1, np.uint8))
    assert h5.from_h5(f, 'a').shape == (3,) and h5.from_h5(f, 'b').shape == (2, 2) and h5.from_h5(f, 'c').shape == (1,)
    with pytest.warns(RuntimeWarning):
        h5.to_h5(f, mode='w', driver='core', a=np.arange(4))
    assert h5.from_h5(f, 'a').shape == (4,)
    with pytest.raises(ValueError):
        h5.to_h5(f, mode='q', a=np.arange(4))


@pytest.mark.skipif(not h5.hdf5_available(), reason='libhdf5 not present')
def test_chunks_and_gzip_compression(tmp_path):
    """``chunks`` / ``compression`` of the reference's to_h5 (util/util.py:1385-1395 -> h5py create_dataset) through libhdf5's
    dataset-creation property list: the data read back are identical, the dataset is chunked, carries one filter and needs
    less storage than the raw array."""
    rng = np.random.default_rng(1)
    labels = np.zeros((600, 500, 2), np.int32)
    labels[100:300, 50:400, 0] = rng.integers(1, 50, (200, 350)) // 10  # a label image: long runs of equal values
    contours = rng.random((1000, 32, 2)).astype(np.float32)
    scores = rng.random(1000).astype(np.float32)
    f = str(tmp_path / 'c.h5')
    h5.to_h5(f, labels=labels, contours=contours, scores=scores, compression='gzip',
             chunks=dict(labels=(128, 128, 1), contours=True, scores=None))
    lay = {k: h5.dataset_layout(f, k) for k in ('labels', 'contours', 'scores')}
    assert lay['labels']['chunks'] == (128, 128, 1) and lay['labels']['filters'] == 1
    assert lay['labels']['storage_bytes'] < labels.nbytes // 20
    # ``chunks=True`` on an N-D array: ``isinstance(True, int)`` holds at util/util.py:1387 -> min(256, extent) per axis
    assert lay['contours']['chunks'] == (256, 32, 2) and lay['contours']['filters'] == 1
    assert lay['scores']['chunks'] == h5.guess_chunk(scores.shape, 4)  # a filter needs chunks: auto-chunked like h5py
    for k, v in (('labels', labels), ('contours', contours), ('scores', scores)):
        got = h5.from_h5(f, k)
        assert got.dtype == v.dtype
        np.testing.assert_array_equal(got, v)
    # an integer chunk setting: the reference turns it into min(256, dim) per dimension for arrays with > 1 dimensions
    h5.to_h5(f, mode='w', labels=labels, scores=scores, chunks=64, compression=9)
    assert h5.dataset_layout(f, 'labels')['chunks'] == (256, 256, 2) and h5.dataset_layout(f, 'scores')['chunks'] == (64,)
    np.testing.assert_array_equal(h5.from_h5(f, 'labels'), labels)
    # plain contiguous datasets stay what they were; empty arrays cannot be chunked; unsupported filters fail loudly
    h5.to_h5(f, mode='w', labels=labels, empty=np.zeros((0, 4), np.float32), compression='gzip', chunks=dict(labels=None, empty=True))
    assert h5.from_h5(f, 'empty').shape == (0, 4) and h5.dataset_layout(f, 'empty')['chunks'] is None
    h5.to_h5(f, mode='w', labels=labels)
    assert h5.dataset_layout(f, 'labels') == dict(chunks=None, filters=0, storage_bytes=labels.nbytes)
    with pytest.raises(NotImplementedError):
        h5.to_h5(f, mode='w', labels=labels, compression='lzf')
    c = h5.guess_chunk((1000, 32, 2), 4)  # 256 KB dataset: ~16 KiB * 2^log10(0.24) = 10 KiB chunks
    assert all(1 <= a <= b for a, b in zip(c, (1000, 32, 2))) and 4 * np.prod(c) <= 16 * 1024


# ---- interchange with h5py-written files -------------------------------------------------------------------
# h5py is absent from the image, so the file below is laid out with RAW libhdf5 calls exactly the way h5py lays out what the
# reference script writes (cpn_inference.py:822-823 -> util/util.py:1385-1399): ``create_dataset(data=..., chunks=...,
# compression='gzip')`` and ``ds.attrs.update({'args': <str>})`` = a variable-length UTF-8 string on a scalar dataspace.
def _h5py_style_file(path, datasets, str_attrs, chunked=()):
    import ctypes
    from ctypes import c_char_p, c_int64, c_uint64, c_void_p
    lib = h5._lib()  # the loaded libhdf5 (ctypes prototypes only; nothing of the product's writer is used below)
    g = lambda n: c_int64.in_dll(lib, n).value
    tids = {'float32': 'H5T_NATIVE_FLOAT_g', 'int64': 'H5T_NATIVE_INT64_g', 'uint8': 'H5T_NATIVE_UINT8_g',
            'float64': 'H5T_NATIVE_DOUBLE_g'}
    f = lib.H5Fcreate(path.encode(), 2, 0, 0)
    assert f >= 0
    for key, arr in datasets.items():
        arr = np.ascontiguousarray(arr)
        space = lib.H5Screate_simple(arr.ndim, (c_uint64 * arr.ndim)(*arr.shape), None)
        dcpl = 0
        if key in chunked:
            dcpl = lib.H5Pcreate(g('H5P_CLS_DATASET_CREATE_ID_g'))
            assert lib.H5Pset_chunk(dcpl, arr.ndim, (c_uint64 * arr.ndim)(*chunked[key])) >= 0
            assert lib.H5Pset_deflate(dcpl, 4) >= 0
        tid = g(tids[arr.dtype.name])
        ds = lib.H5Dcreate2(f, key.encode(), tid, space, 0, dcpl, 0)
        assert ds >= 0
        assert lib.H5Dwrite(ds, tid, 0, 0, 0, arr.ctypes.data_as(c_void_p)) >= 0
        for an, av in str_attrs.get(key, {}).items():
            t = lib.H5Tcopy(g('H5T_C_S1_g'))
            assert lib.H5Tset_size(t, ctypes.c_size_t(-1).value) >= 0  # H5T_VARIABLE
            assert lib.H5Tset_cset(t, 1) >= 0  # H5T_CSET_UTF8
            sp = lib.H5Screate(0)  # H5S_SCALAR
            a = lib.H5Acreate2(ds, an.encode(), t, sp, 0, 0)
            assert a >= 0
            raw = ctypes.create_string_buffer(av.encode('utf-8'))
            ptr = (c_char_p * 1)(ctypes.cast(raw, c_char_p))
            assert lib.H5Awrite(a, t, ptr) >= 0
            lib.H5Aclose(a), lib.H5Sclose(sp), lib.H5Tclose(t)
        lib.H5Dclose(ds), lib.H5Sclose(space)
        if dcpl:
            lib.H5Pclose(dcpl)
    lib.H5Fclose(f)


@pytest.mark.skipif(not h5.hdf5_available(), reason='libhdf5 not present')
def test_reads_a_file_laid_out_like_h5py_writes_it(tmp_path, capsys):
    rng = np.random.default_rng(5)
    k, s = 300, 32
    data = dict(contours=rng.random((k, s, 2)).astype(np.float32), classes=rng.integers(0, 3, k).astype(np.int64),
                scores=rng.random(k).astype(np.float32), labels=(rng.random((90, 70, 2)) > .7).astype(np.uint8))
    args = json.dumps(dict(model='ginoro_CpnResNeXt101UNet-fbe875f1a3e5ce2c', tile_size=1024, stride=768, note='µm – ü'),
                      ensure_ascii=False)
    f = str(tmp_path / 'ref.h5')
    _h5py_style_file(f, data, dict(contours=dict(args=args)), chunked=dict(labels=(32, 32, 1), contours=(64, 32, 2)))
    # the reference's call forms (util/util.py:1459-1488)
    con = h5.from_h5(f, 'contours')
    np.testing.assert_array_equal(con, data['contours'])
    (con, cls, lab), attrs = h5.from_h5(f, 'contours', 'classes', 'labels', return_attrs=True)
    assert cls.dtype == np.int64 and lab.dtype == np.uint8
    np.testing.assert_array_equal(cls, data['classes']), np.testing.assert_array_equal(lab, data['labels'])
    assert attrs[0] == dict(args=args) and isinstance(attrs[0]['args'], str) and attrs[1] == {} and attrs[2] == {}
    assert json.loads(attrs[0]['args'])['note'] == 'µm – ü'
    one, one_attrs = h5.from_h5(f, 'scores', return_attrs=True)  # a single key: the array itself, attrs still a tuple
    assert isinstance(one, np.ndarray) and one_attrs == ({},)
    # **keys_slices: hyperslab reads (chunked + gzip dataset as well), positional keys first
    sc, part, row, lab0 = h5.from_h5(f, 'scores', contours=slice(10, 42), classes=7, labels=(slice(5, 60, 3), Ellipsis, 1))
    np.testing.assert_array_equal(part, data['contours'][10:42])
    assert row == data['classes'][7] and row.shape == ()
    np.testing.assert_array_equal(lab0, data['labels'][5:60:3, ..., 1])
    np.testing.assert_array_equal(h5.from_h5(f, contours=(slice(None), -1, 0)), data['contours'][:, -1, 0])
    np.testing.assert_array_equal(h5.from_h5(f, contours=[3, 1, 2]), data['contours'][[3, 1, 2]])  # in-memory fallback
    np.testing.assert_array_equal(h5.from_h5(f, contours=slice(250, 1000)), data['contours'][250:])
    assert h5.from_h5(f, contours=slice(20, 10)).shape == (0, s, 2)
    with pytest.raises(IndexError):
        h5.from_h5(f, classes=k)
    with pytest.raises(KeyError):
        h5.from_h5(f, 'boxes')
    assert h5.from_h5(f) == () and 'Available keys:' in capsys.readouterr().out
    # `attributes` is no keyword of the reference's from_h5: it names a dataset to slice -> the reference's KeyError
    with pytest.raises(KeyError):
        h5.from_h5(f, 'contours', attributes=True)


@pytest.mark.skipif(not h5.hdf5_available(), reason='libhdf5 not present')
def test_attribute_types_follow_h5py(tmp_path):
    """str -> variable-length UTF-8 (reads back as str), bytes -> variable-length ASCII (str), numpy.bytes_ -> fixed length
    (numpy.bytes_), numbers / arrays -> native numeric types, bool -> the {FALSE, TRUE} enum, dict -> its JSON text."""
    import ctypes
    f = str(tmp_path / 'a.h5')
    at = dict(args=json.dumps(dict(a=1, b=None)), unicode='Zellkörper ∆', empty='', raw=b'ascii bytes', fixed=np.bytes_(b'fixed'),
              n=7, x=0.25, f32=np.float32(1.5), u8=np.uint8(200), flag=True, arr=np.arange(6, dtype=np.int32).reshape(2, 3),
              spacing=[0.5, 0.5], names=['a', 'bcd', 'ü'], cfg=dict(tile=512), t=torch.arange(3))
    h5.to_h5(f, contours=np.zeros((2, 4, 2), np.float32), scores=np.zeros(2, np.float32), attributes=dict(contours=at))
    _, (got, none) = h5.from_h5(f, 'contours', 'scores', return_attrs=True)
    assert none == {} and set(got) == set(at)
    for k in ('args', 'unicode', 'empty'):
        assert type(got[k]) is str and got[k] == at[k]
    assert got['raw'] == 'ascii bytes' and type(got['fixed']) is np.bytes_ and got['fixed'] == b'fixed'
    assert got['n'] == 7 and got['n'].dtype == np.int64 and got['x'] == .25 and got['x'].dtype == np.float64
    assert got['f32'].dtype == np.float32 and got['u8'].dtype == np.uint8 and got['u8'] == 200
    assert got['flag'].dtype == np.bool_ and bool(got['flag']) is True
    np.testing.assert_array_equal(got['arr'], at['arr']) and got['arr'].dtype == np.int32
    np.testing.assert_array_equal(got['spacing'], [.5, .5])
    assert list(got['names']) == ['a', 'bcd', 'ü'] and json.loads(got['cfg']) == dict(tile=512)
    np.testing.assert_array_equal(got['t'], [0, 1, 2])
    # what is on disk for ``args``: a variable-length, UTF-8 string on a scalar dataspace (what h5py reads as ``str``)
    lib = h5._lib()
    fh = lib.H5Fopen(f.encode(), 0, 0)
    ds = lib.H5Dopen2(fh, b'contours', 0)
    a = lib.H5Aopen(ds, b'args', 0)
    t, sp = lib.H5Aget_type(a), lib.H5Aget_space(a)
    assert lib.H5Tget_class(t) == 3 and lib.H5Tis_variable_str(t) > 0 and lib.H5Tget_cset(t) == 1
    assert lib.H5Sget_simple_extent_ndims(sp) == 0
    lib.H5Sclose(sp), lib.H5Tclose(t), lib.H5Aclose(a), lib.H5Dclose(ds), lib.H5Fclose(fh)
    # attrs.update on an existing dataset replaces the value; the dataset keeps its contents when written in place
    h5.to_h5(f, mode='a', contours=np.ones((2, 4, 2), np.float32), attributes=dict(contours=dict(args='second', n=8)))
    con, (got2,) = h5.from_h5(f, 'contours', return_attrs=True)
    assert got2['args'] == 'second' and got2['n'] == 8 and got2['unicode'] == at['unicode'] and con.min() == 1
    with pytest.raises(TypeError):
        h5.to_h5(f, mode='a', scores=np.zeros(2, np.float32), attributes=dict(scores=dict(bad=None)))
