"""
Keras `.h5` checkpoints without h5py: a ctypes binding of the few HDF5 C-library calls that reading (and, for
fixtures and round trips, writing) a Keras weight file needs.

The reference loads and saves its models with `keras.models.load_model` / `model.save`
(keras_retinanet_3D/models/__init__.py:81, bin/convert_model.py:50-53); the file those produce is HDF5 in the
layout of keras/engine/saving.py (Keras 2.2):

    /                                 attrs keras_version, backend, model_config (JSON), training_config
    /model_weights                    attrs layer_names = [b'conv1', b'bn_conv1', ..., b'regression_submodel', ...]
    /model_weights/<layer>            attrs weight_names = [b'conv1/kernel:0', ...]
    /model_weights/<layer>/<weight>   float32 dataset; <weight> = '<variable scope>/<variable>:0', i.e. the dataset of
                                      conv1's kernel is /model_weights/conv1/conv1/kernel:0 and a layer of a nested
                                      sub-model sits at /model_weights/regression_submodel/pyramid_regression_0/kernel:0
    (a weights-only file from `model.save_weights` has the same groups directly under /)

h5py is preferred when it is importable; otherwise libhdf5 itself is bound (`GPP_HDF5_LIB`, the loader's search path, or the
copies a conda / distribution install leaves behind).  HDF5 1.10 and 1.12+ are both handled (`hid_t` is 64-bit in both).
"""
import ctypes
import ctypes.util
import glob
import os

import numpy as np

_H5F_ACC_RDONLY, _H5F_ACC_TRUNC = 0, 2
_H5P_DEFAULT, _H5S_ALL = 0, 0
_H5I_GROUP, _H5I_DATASET = 2, 5
_H5T_INTEGER, _H5T_FLOAT, _H5T_STRING = 0, 1, 3
_H5T_STR_NULLPAD = 1

hid_t = ctypes.c_int64


class Hdf5Error(IOError):
    pass


_lib = None


def _candidates():
    env = os.environ.get('GPP_HDF5_LIB')
    if env:
        yield env
    found = ctypes.util.find_library('hdf5') or ctypes.util.find_library('hdf5_serial')
    if found:
        yield found
    for pat in ('/usr/lib/x86_64-linux-gnu/libhdf5_serial.so*', '/usr/lib/x86_64-linux-gnu/libhdf5.so*', '/usr/lib64/libhdf5.so*',
                '/usr/local/lib/libhdf5.so*', '/opt/conda/lib/libhdf5.so*', os.path.expanduser('~/miniconda3/lib/libhdf5.so*')):
        for p in sorted(glob.glob(pat)):
            yield p


def library():
    """ the bound libhdf5 (loaded once); Hdf5Error when none can be found """
    global _lib
    if _lib is not None:
        return _lib
    tried = []
    for cand in _candidates():
        try:
            lib = ctypes.CDLL(cand)
        except OSError as e:
            tried.append('{} ({})'.format(cand, e))
            continue
        if not hasattr(lib, 'H5Fopen'):
            continue
        _declare(lib)
        if lib.H5open() < 0:
            raise Hdf5Error('H5open failed in {}'.format(cand))
        lib.H5Eset_auto2(hid_t(0), None, None)        # errors are reported through return codes, not printed stacks
        _lib = lib
        return lib
    raise Hdf5Error('reading a Keras .h5 file needs h5py or an HDF5 C library; none found (set GPP_HDF5_LIB=/path/to/libhdf5.so'
                    '{})'.format('; tried ' + ', '.join(tried) if tried else ''))


H5L_iterate_t = ctypes.CFUNCTYPE(ctypes.c_int, hid_t, ctypes.c_char_p, ctypes.c_void_p, ctypes.c_void_p)


def _declare(lib):
    P, I, S, U = ctypes.c_void_p, ctypes.c_int, ctypes.c_char_p, ctypes.c_uint
    sig = {
        'H5open': (I, []), 'H5Eset_auto2': (I, [hid_t, P, P]),
        'H5Fopen': (hid_t, [S, U, hid_t]), 'H5Fcreate': (hid_t, [S, U, hid_t, hid_t]), 'H5Fclose': (I, [hid_t]),
        'H5Fis_hdf5': (I, [S]),
        'H5Gcreate2': (hid_t, [hid_t, S, hid_t, hid_t, hid_t]), 'H5Gclose': (I, [hid_t]),
        'H5Oopen': (hid_t, [hid_t, S, hid_t]), 'H5Oclose': (I, [hid_t]), 'H5Iget_type': (I, [hid_t]),
        'H5Lexists': (I, [hid_t, S, hid_t]),
        'H5Dopen2': (hid_t, [hid_t, S, hid_t]), 'H5Dclose': (I, [hid_t]), 'H5Dget_space': (hid_t, [hid_t]),
        'H5Dget_type': (hid_t, [hid_t]), 'H5Dread': (I, [hid_t, hid_t, hid_t, hid_t, hid_t, P]),
        'H5Dcreate2': (hid_t, [hid_t, S, hid_t, hid_t, hid_t, hid_t, hid_t]), 'H5Dwrite': (I, [hid_t, hid_t, hid_t, hid_t, hid_t, P]),
        'H5Dvlen_reclaim': (I, [hid_t, hid_t, hid_t, P]),
        'H5Sget_simple_extent_ndims': (I, [hid_t]), 'H5Sget_simple_extent_dims': (I, [hid_t, P, P]),
        'H5Sget_simple_extent_npoints': (ctypes.c_int64, [hid_t]), 'H5Sclose': (I, [hid_t]),
        'H5Screate_simple': (hid_t, [I, P, P]), 'H5Screate': (hid_t, [I]),
        'H5Tget_class': (I, [hid_t]), 'H5Tget_size': (ctypes.c_size_t, [hid_t]), 'H5Tclose': (I, [hid_t]),
        'H5Tis_variable_str': (I, [hid_t]), 'H5Tcopy': (hid_t, [hid_t]), 'H5Tset_size': (I, [hid_t, ctypes.c_size_t]),
        'H5Tset_strpad': (I, [hid_t, I]),
        'H5Aexists': (I, [hid_t, S]), 'H5Aopen': (hid_t, [hid_t, S, hid_t]), 'H5Aclose': (I, [hid_t]),
        'H5Aget_type': (hid_t, [hid_t]), 'H5Aget_space': (hid_t, [hid_t]), 'H5Aread': (I, [hid_t, hid_t, P]),
        'H5Acreate2': (hid_t, [hid_t, S, hid_t, hid_t, hid_t, hid_t]), 'H5Awrite': (I, [hid_t, hid_t, P]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args
    # H5Lvisit became a macro over H5Lvisit1 / H5Lvisit2 in HDF5 1.12; the callback's (group, name, info, data) shape is the same
    visit = getattr(lib, 'H5Lvisit', None) or getattr(lib, 'H5Lvisit1')
    visit.restype, visit.argtypes = I, [hid_t, I, I, H5L_iterate_t, P]
    lib._gpp_visit = visit


def _native(lib, name):
    return hid_t.in_dll(lib, name).value


class File(object):
    """ `with File(path) as f:` -- read access: f.datasets(group) -> {path below group: float32 array}, f.attr(path, name),
    f.exists(path); `File(path, 'w')` adds create_group / write_dataset / write_attr (what a Keras-layout fixture needs). """

    def __init__(self, path, mode='r'):
        self.lib = library()
        self.path = path
        bpath = os.fsencode(path)
        if mode == 'r':
            if not os.path.isfile(path):
                raise Hdf5Error('{}: no such file'.format(path))
            if self.lib.H5Fis_hdf5(bpath) <= 0:
                raise Hdf5Error('{}: not an HDF5 file'.format(path))
            self.id = self.lib.H5Fopen(bpath, _H5F_ACC_RDONLY, _H5P_DEFAULT)
        elif mode == 'w':
            self.id = self.lib.H5Fcreate(bpath, _H5F_ACC_TRUNC, _H5P_DEFAULT, _H5P_DEFAULT)
        else:
            raise ValueError("mode must be 'r' or 'w'")
        if self.id < 0:
            raise Hdf5Error('{}: cannot open (mode {})'.format(path, mode))

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def close(self):
        if self.id >= 0:
            self.lib.H5Fclose(self.id)
            self.id = -1

    # ---- reading ----------------------------------------------------------------------------------------------------
    def exists(self, path):
        """ every component of `path` is a link that resolves """
        at = ''
        for part in [p for p in path.split('/') if p]:
            at = at + '/' + part
            if self.lib.H5Lexists(self.id, at.encode(), _H5P_DEFAULT) <= 0:
                return False
        return True

    def links(self, group='/'):
        """ every link path below `group` (recursive), in name order """
        names = []

        def cb(_gid, name, _info, _data):
            names.append(name.decode())
            return 0

        gid = self.lib.H5Oopen(self.id, group.encode(), _H5P_DEFAULT)
        if gid < 0:
            raise Hdf5Error('{}: no object {}'.format(self.path, group))
        try:
            if self.lib._gpp_visit(gid, 0, 0, H5L_iterate_t(cb), None) < 0:       # H5_INDEX_NAME, H5_ITER_INC
                raise Hdf5Error('{}: cannot walk {}'.format(self.path, group))
        finally:
            self.lib.H5Oclose(gid)
        return names

    def read(self, path):
        """ the dataset at `path` as a float32 array (HDF5 converts from the stored float / integer type) """
        lib = self.lib
        did = lib.H5Dopen2(self.id, path.encode(), _H5P_DEFAULT)
        if did < 0:
            raise Hdf5Error('{}: no dataset {}'.format(self.path, path))
        try:
            tid = lib.H5Dget_type(did)
            cls = lib.H5Tget_class(tid)
            lib.H5Tclose(tid)
            if cls not in (_H5T_FLOAT, _H5T_INTEGER):
                raise Hdf5Error('{}: dataset {} is not numeric (HDF5 type class {})'.format(self.path, path, cls))
            sid = lib.H5Dget_space(did)
            nd = lib.H5Sget_simple_extent_ndims(sid)
            dims = (ctypes.c_uint64 * max(nd, 1))()
            if nd > 0:
                lib.H5Sget_simple_extent_dims(sid, dims, None)
            lib.H5Sclose(sid)
            out = np.empty(tuple(int(d) for d in dims[:nd]), dtype=np.float32)
            if out.size and lib.H5Dread(did, _native(lib, 'H5T_NATIVE_FLOAT_g'), _H5S_ALL, _H5S_ALL, _H5P_DEFAULT,
                                        out.ctypes.data_as(ctypes.c_void_p)) < 0:
                raise Hdf5Error('{}: cannot read dataset {}'.format(self.path, path))
            return out
        finally:
            lib.H5Dclose(did)

    def datasets(self, group='/'):
        """ {path relative to `group`: float32 array} of every numeric dataset below `group` """
        lib = self.lib
        out = {}
        base = group.rstrip('/')
        for name in self.links(group):
            full = base + '/' + name
            oid = lib.H5Oopen(self.id, full.encode(), _H5P_DEFAULT)
            if oid < 0:
                continue                                        # dangling soft / external link
            kind = lib.H5Iget_type(oid)
            lib.H5Oclose(oid)
            if kind == _H5I_DATASET:
                out[name] = self.read(full)
        return out

    def attr(self, path, name):
        """ attribute `name` of the object at `path`: a list of bytes for string attributes (fixed-length as h5py writes NumPy
        'S' arrays, or variable-length), a float64 array for numeric ones; None when the attribute does not exist """
        lib = self.lib
        oid = lib.H5Oopen(self.id, path.encode(), _H5P_DEFAULT)
        if oid < 0:
            raise Hdf5Error('{}: no object {}'.format(self.path, path))
        try:
            if lib.H5Aexists(oid, name.encode()) <= 0:
                return None
            aid = lib.H5Aopen(oid, name.encode(), _H5P_DEFAULT)
            tid = lib.H5Aget_type(aid)
            sid = lib.H5Aget_space(aid)
            try:
                n = int(lib.H5Sget_simple_extent_npoints(sid))
                cls = lib.H5Tget_class(tid)
                if cls == _H5T_STRING:
                    if lib.H5Tis_variable_str(tid) > 0:
                        buf = (ctypes.c_char_p * n)()
                        if lib.H5Aread(aid, tid, buf) < 0:
                            raise Hdf5Error('{}: cannot read attribute {} of {}'.format(self.path, name, path))
                        vals = [bytes(b) if b is not None else b'' for b in buf]
                        lib.H5Dvlen_reclaim(tid, sid, _H5P_DEFAULT, buf)
                        return vals
                    size = int(lib.H5Tget_size(tid))
                    raw = ctypes.create_string_buffer(n * size)
                    if lib.H5Aread(aid, tid, raw) < 0:
                        raise Hdf5Error('{}: cannot read attribute {} of {}'.format(self.path, name, path))
                    return [raw.raw[i * size:(i + 1) * size].split(b'\0', 1)[0] for i in range(n)]
                vals = np.empty(n, np.float64)
                if lib.H5Aread(aid, _native(lib, 'H5T_NATIVE_DOUBLE_g'), vals.ctypes.data_as(ctypes.c_void_p)) < 0:
                    raise Hdf5Error('{}: cannot read attribute {} of {}'.format(self.path, name, path))
                return vals
            finally:
                lib.H5Sclose(sid)
                lib.H5Tclose(tid)
                lib.H5Aclose(aid)
        finally:
            lib.H5Oclose(oid)

    # ---- writing (fixtures, round trips) ---------------------------------------------------------------------------
    def create_group(self, path):
        at = ''
        for part in [p for p in path.split('/') if p]:
            at = at + '/' + part
            if self.lib.H5Lexists(self.id, at.encode(), _H5P_DEFAULT) > 0:
                continue
            gid = self.lib.H5Gcreate2(self.id, at.encode(), _H5P_DEFAULT, _H5P_DEFAULT, _H5P_DEFAULT)
            if gid < 0:
                raise Hdf5Error('{}: cannot create group {}'.format(self.path, at))
            self.lib.H5Gclose(gid)

    def write_dataset(self, path, array):
        """ float32, contiguous storage (what Keras writes for weights) """
        lib = self.lib
        a = np.ascontiguousarray(array, dtype=np.float32)
        parent = path.rsplit('/', 1)[0]
        if parent:
            self.create_group(parent)
        dims = (ctypes.c_uint64 * max(a.ndim, 1))(*a.shape)
        sid = lib.H5Screate_simple(a.ndim, dims, None) if a.ndim else lib.H5Screate(0)      # H5S_SCALAR
        did = lib.H5Dcreate2(self.id, path.encode(), _native(lib, 'H5T_IEEE_F32LE_g'), sid, _H5P_DEFAULT, _H5P_DEFAULT, _H5P_DEFAULT)
        try:
            if did < 0 or (a.size and lib.H5Dwrite(did, _native(lib, 'H5T_NATIVE_FLOAT_g'), _H5S_ALL, _H5S_ALL, _H5P_DEFAULT,
                                                   a.ctypes.data_as(ctypes.c_void_p)) < 0):
                raise Hdf5Error('{}: cannot write dataset {}'.format(self.path, path))
        finally:
            if did >= 0:
                lib.H5Dclose(did)
            lib.H5Sclose(sid)

    def write_attr(self, path, name, values):
        """ a list of bytes -> a 1-D array of fixed-length, null-padded strings (h5py's encoding of a NumPy 'S' array: how Keras
        stores layer_names / weight_names); a single bytes object -> a scalar string attribute """
        lib = self.lib
        scalar = isinstance(values, (bytes, str))
        vals = [values] if scalar else list(values)
        vals = [v.encode() if isinstance(v, str) else bytes(v) for v in vals]
        size = max([len(v) for v in vals] + [1])
        oid = lib.H5Oopen(self.id, path.encode(), _H5P_DEFAULT)
        if oid < 0:
            raise Hdf5Error('{}: no object {}'.format(self.path, path))
        tid = lib.H5Tcopy(_native(lib, 'H5T_C_S1_g'))
        lib.H5Tset_size(tid, size)
        lib.H5Tset_strpad(tid, _H5T_STR_NULLPAD)
        dims = (ctypes.c_uint64 * 1)(len(vals))
        sid = lib.H5Screate(0) if scalar else lib.H5Screate_simple(1, dims, None)
        aid = lib.H5Acreate2(oid, name.encode(), tid, sid, _H5P_DEFAULT, _H5P_DEFAULT)
        try:
            raw = b''.join(v.ljust(size, b'\0') for v in vals)
            if aid < 0 or (vals and lib.H5Awrite(aid, tid, ctypes.create_string_buffer(raw, len(raw))) < 0):
                raise Hdf5Error('{}: cannot write attribute {} of {}'.format(self.path, name, path))
        finally:
            if aid >= 0:
                lib.H5Aclose(aid)
            lib.H5Sclose(sid)
            lib.H5Tclose(tid)
            lib.H5Oclose(oid)
