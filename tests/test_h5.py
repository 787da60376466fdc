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
    *arrays, attrs = h5.from_h5(f, *names, attributes=True)
    for name, a in zip(names, arrays):
        exp = out[name].numpy() if isinstance(out[name], torch.Tensor) else out[name]
        assert a.dtype == exp.dtype and a.shape == exp.shape, name
        np.testing.assert_array_equal(a, exp)
    assert json.loads(attrs['contours']['args']) == args
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
    assert lay['contours']['chunks'] == h5.guess_chunk(contours.shape, 4) and lay['contours']['filters'] == 1
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
