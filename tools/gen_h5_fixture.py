#!/usr/bin/env python3
"""
Writes tests/golden/keras_layout_h5import.h5 with the HDF5 project's own command-line tool `h5import` (found on PATH or in
/opt/conda/bin): a file in the group / dataset layout of a Keras `model.save` checkpoint that neither h5py nor this repository's
ctypes writer (models/hdf5.py) has touched, so that tests/test_weights_h5.py reads bytes laid out by the real HDF5 library.
`h5import` cannot write attributes: this fixture therefore has no layer_names / weight_names and exercises the dataset-walk
branch of load_keras_h5; the attribute branch is tested on files from models/hdf5.py's writer, cross-checked with `h5dump`.

The values are a fixed function of the dataset name (see `values`), which the test recomputes.
One dataset is stored chunked + gzip-compressed and one as float64, as a checkpoint re-packed by h5repack might hold them.
"""
import os
import shutil
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, 'tests', 'golden', 'keras_layout_h5import.h5')

DATASETS = [   # (path, shape, extra h5import configuration lines)
    ('model_weights/conv1/conv1/kernel:0', (7, 7, 3, 4), []),
    ('model_weights/bn_conv1/bn_conv1/gamma:0', (4,), []),
    ('model_weights/bn_conv1/bn_conv1/beta:0', (4,), []),
    ('model_weights/bn_conv1/bn_conv1/moving_mean:0', (4,), []),
    ('model_weights/bn_conv1/bn_conv1/moving_variance:0', (4,), ['OUTPUT-SIZE 64']),
    ('model_weights/regression_submodel/pyramid_regression_0/kernel:0', (3, 3, 8, 8),
     ['CHUNKED-DIMENSION-SIZES 3 3 4 4', 'COMPRESSION-TYPE GZIP', 'COMPRESSION-PARAM 6']),
    ('model_weights/regression_submodel/pyramid_regression_0/bias:0', (8,), []),
    ('model_weights/classification_submodel/pyramid_classification/bias:0', (96,), []),
]


def values(path, shape):
    """ dyadic rationals (exact in text, float32 and float64) that depend on the name and the position """
    n = int(np.prod(shape))
    seed = sum(path.encode()) % 97
    return (((np.arange(n) * 37 + seed) % 1024) - 512).astype(np.float64).reshape(shape) / 64.0


def main():
    tool = shutil.which('h5import') or '/opt/conda/bin/h5import'
    if not os.path.isfile(tool):
        sys.exit('h5import not found')
    tmp = tempfile.mkdtemp()
    out = os.path.join(tmp, 'out.h5')
    for i, (path, shape, extra) in enumerate(DATASETS):
        txt, cfg = os.path.join(tmp, 'd{}.txt'.format(i)), os.path.join(tmp, 'd{}.cfg'.format(i))
        with open(txt, 'w') as f:
            f.write(' '.join(repr(float(v)) for v in values(path, shape).ravel()) + '\n')
        lines = ['PATH ' + path, 'INPUT-CLASS TEXTFP', 'RANK {}'.format(len(shape)),
                 'DIMENSION-SIZES ' + ' '.join(str(s) for s in shape), 'OUTPUT-CLASS FP']
        if not any(e.startswith('OUTPUT-SIZE') for e in extra):
            lines.append('OUTPUT-SIZE 32')
        with open(cfg, 'w') as f:
            f.write('\n'.join(lines + extra) + '\n')
        subprocess.check_call([tool, txt, '-c', cfg, '-o', out])
    shutil.copyfile(out, OUT)
    shutil.rmtree(tmp)
    print('wrote', OUT, os.path.getsize(OUT), 'bytes')


if __name__ == '__main__':
    main()
