"""
Row f2: the Keras `.h5` checkpoint the reference reads and writes (models/__init__.py:81,
bin/convert_model.py:50-53).  No h5py in this image, so models/hdf5.py binds libhdf5 itself; these tests run that
branch against (a) a fixture laid out by the HDF5 project's own `h5import` tool (tools/gen_h5_fixture.py), (b) files from
this repository's writer, cross-checked with `h5dump` where that tool exists.
"""
import importlib.util
import os
import shutil
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'ground-plane-polling_amd'))

from keras_retinanet_3D.models import hdf5, weights as W  # noqa: E402

try:
    hdf5.library()
    HAVE_HDF5 = True
except hdf5.Hdf5Error:
    HAVE_HDF5 = False

needs_hdf5 = pytest.mark.skipif(not HAVE_HDF5 and importlib.util.find_spec('h5py') is None,
                                reason='neither libhdf5 nor h5py on this machine')
FIXTURE = os.path.join(ROOT, 'tests', 'golden', 'keras_layout_h5import.h5')


def _gen():
    spec = importlib.util.spec_from_file_location('gen_h5_fixture', os.path.join(ROOT, 'tools', 'gen_h5_fixture.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@needs_hdf5
def test_reads_a_file_laid_out_by_the_hdf5_tools():
    gen = _gen()
    got = W.load_weights(FIXTURE)
    assert sorted(got) == sorted(W._array_key(p) for p, _, _ in gen.DATASETS)
    for path, shape, _ in gen.DATASETS:             # contiguous float32, float64, chunked + gzip: all come back as float32
        arr = got[W._array_key(path)]
        assert arr.dtype == np.float32 and arr.shape == shape
        assert np.array_equal(arr, gen.values(path, shape).astype(np.float32)), path


@needs_hdf5
@pytest.mark.parametrize('backbone', ['resnet50', 'resnet101'])
def test_keras_layout_round_trip_of_a_whole_model(tmp_path, backbone):
    w = W.synthetic_weights(backbone, 5)
    path = str(tmp_path / 'model.h5')
    W.save_weights(path, w)
    back = W.load_weights(path)
    W.validate_weights(back, backbone)
    assert sorted(back) == sorted(w)
    assert all(np.array_equal(back[k], w[k]) and back[k].dtype == np.float32 for k in w)
    with hdf5.File(path) as f:                       # the attributes Keras' loader walks, nested sub-models included
        layers = [b.decode() for b in f.attr('/model_weights', 'layer_names')]
        assert 'conv1' in layers and 'regression_submodel' in layers and 'pyramid_regression_0' not in layers
        names = [b.decode() for b in f.attr('/model_weights/regression_submodel', 'weight_names')]
        assert 'pyramid_regression_0/kernel:0' in names and 'pyramid_regression_op5/bias:0' in names
        assert not any(n.startswith('pyramid_regression_dim') for n in names)
        assert f.attr('/', 'keras_version') == [b'2.2.0'] and f.attr('/', 'nope') is None
        assert f.exists('/model_weights/bn_conv1/bn_conv1/moving_variance:0') and not f.exists('/model_weights/conv2')


@needs_hdf5
def test_weights_only_file_and_files_without_attributes(tmp_path):
    """ `model.save_weights` puts the layer groups at the root; a re-packed file may have lost the attributes """
    path = str(tmp_path / 'weights_only.h5')
    a = np.arange(24, dtype=np.float32).reshape(1, 1, 4, 6)
    with hdf5.File(path, 'w') as f:
        f.write_dataset('/C5_reduced/C5_reduced/kernel:0', a)
        f.write_dataset('/C5_reduced/C5_reduced/bias:0', a[0, 0, 0])
        f.write_attr('/', 'layer_names', [b'C5_reduced'])
        f.write_attr('/C5_reduced', 'weight_names', [b'C5_reduced/kernel:0', b'C5_reduced/bias:0'])
    got = W.load_weights(path)
    assert sorted(got) == ['C5_reduced/bias', 'C5_reduced/kernel'] and np.array_equal(got['C5_reduced/kernel'], a)
    bare = str(tmp_path / 'bare.hdf5')
    with hdf5.File(bare, 'w') as f:
        f.write_dataset('/model_weights/P5/P5/kernel:0', a)
        f.write_dataset('/optimizer_weights/Adam/iterations:0', np.float32(3))       # outside model_weights: ignored
    got = W.load_weights(bare)
    assert list(got) == ['P5/kernel'] and np.array_equal(got['P5/kernel'], a)


@needs_hdf5
def test_errors_name_the_problem(tmp_path):
    dup = str(tmp_path / 'dup.h5')
    with hdf5.File(dup, 'w') as f:
        f.write_dataset('/model_weights/a/P5/kernel:0', np.zeros(3, np.float32))
        f.write_dataset('/model_weights/b/P5/kernel:0', np.zeros(3, np.float32))
    with pytest.raises(ValueError, match='appears twice'):
        W.load_weights(dup)
    notes = str(tmp_path / 'not_hdf5.h5')
    with open(notes, 'wb') as f:
        f.write(b'PK\x03\x04 this is not HDF5')
    with pytest.raises(IOError):
        W.load_weights(notes)
    with pytest.raises(IOError):
        W.load_weights(str(tmp_path / 'missing.h5'))
    part = str(tmp_path / 'part.h5')                   # a checkpoint of another architecture: one clear error at load_model time
    w = W.synthetic_weights('resnet50', 1)
    del w['res4c_branch2b/kernel']
    w['P3/kernel'] = w['P3/kernel'].transpose(3, 2, 0, 1)
    W.save_weights(part, w)
    with pytest.raises(ValueError, match=r'1 arrays missing: res4c_branch2b/kernel; 1 with a wrong shape: P3/kernel'):
        W.validate_weights(W.load_weights(part), 'resnet50')


@needs_hdf5
@pytest.mark.skipif(not (shutil.which('h5dump') or os.path.isfile('/opt/conda/bin/h5dump')), reason='h5dump not installed')
def test_the_writer_is_read_back_by_h5dump(tmp_path):
    """ the HDF5 project's own dump tool sees the Keras layout in a file from models/hdf5.py's writer """
    tool = shutil.which('h5dump') or '/opt/conda/bin/h5dump'
    path = str(tmp_path / 'm.h5')
    W.save_weights(path, {'conv1/kernel': np.array([[1.5, -2.25], [3.0, 4.0]], np.float32),
                          'pyramid_regression_dim/bias': np.array([0.5, 0.25, 8.0], np.float32)})
    dump = subprocess.check_output([tool, path]).decode()
    flat = ' '.join(dump.split())
    assert 'GROUP "model_weights"' in flat and 'GROUP "regression_dim_submodel"' in flat and 'DATASET "kernel:0"' in flat
    assert 'ATTRIBUTE "layer_names"' in flat and '(0): "conv1\\000' in flat and '(1): "regression_dim_submodel"' in flat
    assert 'ATTRIBUTE "weight_names"' in flat and '"pyramid_regression_dim/bias:0"' in flat
    assert 'H5T_IEEE_F32LE' in flat and '1.5, -2.25' in flat and '0.5, 0.25, 8' in flat
    assert 'STRPAD H5T_STR_NULLPAD' in flat            # h5py's encoding of NumPy 'S' arrays


def test_load_model_accepts_an_h5_path_in_principle():
    """ the extension dispatch itself needs no HDF5 library """
    with pytest.raises(ValueError, match='unknown weight file type'):
        W.load_weights('model.ckpt')
