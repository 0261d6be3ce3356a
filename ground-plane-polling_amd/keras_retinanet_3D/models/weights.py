"""
Weights of the RetinaNet-3D graph: layer inventory, synthetic (seeded) initialisation, files,
and folding of the frozen BatchNormalization layers into the preceding convolutions.

The layer names are the Keras names of the reference graph, so that a converted `.h5`
(reference bin/convert_model.py:43-53) maps one to one onto this container:
    FPN     C5_reduced P5 C4_reduced P4 C3_reduced P3 P6 P7          models/retinanet.py:183-203
    heads   pyramid_regression_{0..3}, pyramid_regression_op{1..5}   models/retinanet.py:100-120
            pyramid_regression_dim_{0..3}, pyramid_regression_dim    models/retinanet.py:151-161
            pyramid_classification_{0..3}, pyramid_classification    models/retinanet.py:52-69
    backbone (third-party keras_resnet, models/resnet.py:88-93): conv1 / bn_conv1,
            res{S}{B}_branch{2a,2b,2c,1} / bn{S}{B}_branch{...}; block letters a, b, c ... except
            stages 3 and 4 of ResNet-101/152 which use a, b1, b2, ... (keras_resnet numerical names)

Arrays are stored in Keras conventions: conv kernels HWIO float32 '<layer>/kernel', biases
'<layer>/bias', BatchNormalization '<bn>/gamma', '/beta', '/moving_mean', '/moving_variance'.

No trained weights ship with the reference (README.md:75 is a link), so benchmarks and tests use
`synthetic:<seed>` weights: the architecture is exact, the values are random but scaled so that
activations stay O(1) through the 50-152 layers and the number of anchors above the 0.05 score
threshold is realistic (about a thousand per 402x1333 image).
"""

import zlib

import numpy as np

BN_EPSILON = 1e-5                     # keras_resnet BatchNormalization(epsilon=1e-5)
NUM_ANCHORS = 12                      # models/retinanet.py:230-235: 3 ratios x 4 scales
BLOCKS = {'resnet50': (3, 4, 6, 3), 'resnet101': (3, 4, 23, 3), 'resnet152': (3, 8, 36, 3)}
NUMERICAL_NAMES = {'resnet50': (False, False, False, False), 'resnet101': (False, True, True, False),
                   'resnet152': (False, True, True, False)}


def block_name(backbone, stage, block):
    """ keras_resnet naming: stage 0..3 -> '2'..'5'; block letter or 'b<k>' """
    if block > 0 and NUMERICAL_NAMES[backbone][stage]:
        return '{}b{}'.format(stage + 2, block)
    return '{}{}'.format(stage + 2, chr(ord('a') + block))


def backbone_layers(backbone):
    """ [(conv name, bn name, KH, KW, C_in, C_out, stride)] of the ResNet, in execution order """
    layers = [('conv1', 'bn_conv1', 7, 7, 3, 64, 2)]
    c_in = 64
    for stage, n_blocks in enumerate(BLOCKS[backbone]):
        f = 64 * 2 ** stage
        for block in range(n_blocks):
            nm = block_name(backbone, stage, block)
            stride = 2 if (block == 0 and stage > 0) else 1
            layers.append(('res{}_branch2a'.format(nm), 'bn{}_branch2a'.format(nm), 1, 1, c_in, f, stride))
            layers.append(('res{}_branch2b'.format(nm), 'bn{}_branch2b'.format(nm), 3, 3, f, f, 1))
            layers.append(('res{}_branch2c'.format(nm), 'bn{}_branch2c'.format(nm), 1, 1, f, 4 * f, 1))
            if block == 0:
                layers.append(('res{}_branch1'.format(nm), 'bn{}_branch1'.format(nm), 1, 1, c_in, 4 * f, stride))
            c_in = 4 * f
    return layers


def fpn_layers():
    """ [(name, K, C_in, C_out, stride)] models/retinanet.py:170-205 (feature_size = 512) """
    return [('C5_reduced', 1, 2048, 512, 1), ('P5', 3, 512, 512, 1), ('C4_reduced', 1, 1024, 512, 1), ('P4', 3, 512, 512, 1),
            ('C3_reduced', 1, 512, 512, 1), ('P3', 3, 512, 512, 1), ('P6', 3, 2048, 512, 2), ('P7', 3, 512, 512, 2)]


def head_layers():
    """ [(name, C_in, C_out, kind)] all 3x3 stride 1 'same'; kind = 'tower' (ReLU) | 'out' """
    A = NUM_ANCHORS
    out = []
    for i in range(4):
        out.append(('pyramid_regression_{}'.format(i), 512, 512, 'tower'))
    out += [('pyramid_regression_op1', 512, 4 * A, 'out')] + [('pyramid_regression_op{}'.format(k), 512, 2 * A, 'out') for k in (2, 3, 4, 5)]
    for i in range(4):
        out.append(('pyramid_regression_dim_{}'.format(i), 512 if i == 0 else 128, 128, 'tower'))
    out.append(('pyramid_regression_dim', 128, 3 * A, 'out'))
    for i in range(4):
        out.append(('pyramid_classification_{}'.format(i), 512 if i == 0 else 256, 256, 'tower'))
    out.append(('pyramid_classification', 256, 8 * A, 'out'))
    return out


def _rng(seed, name):
    return np.random.default_rng([int(seed), zlib.crc32(name.encode())])


def _normal(seed, name, shape, std):
    return (_rng(seed, name).standard_normal(size=shape, dtype=np.float32) * np.float32(std)).astype(np.float32)


# calibration of the synthetic output layers (measured once with the float32 CPU oracle on a
# 402x1333 uniform-noise frame; see tests/test_network_oracle.py::test_synthetic_statistics)
SYN_CLS_OUT_GAIN = 0.182      # classification logits ~ N(-4.6, 0.52): about 1e3 anchors of 137256 above 0.05
SYN_REG_OUT_GAIN = 0.31       # regression deltas ~ N(0, 1)
SYN_DIM_OUT_GAIN = 0.18       # dimension deltas ~ N(0, 1)
# the deeper backbones end with somewhat larger pyramid features: same calibration target, measured the same way
SYN_BACKBONE_OUT_SCALE = {'resnet50': 1.0, 'resnet101': 0.676, 'resnet152': 0.52}


def synthetic_weights(backbone='resnet50', seed=1234, family='he'):
    """ Seeded random weights with the exact architecture of `backbone` + FPN + heads.
    family 'he': He-normal kernels, BatchNormalization close to the identity (every activation O(1)).
    family 'trained': the same draw re-parameterised the way a trained checkpoint looks (trained_like below). """
    if family not in ('he', 'trained'):
        raise ValueError("family must be 'he' or 'trained', got {!r}".format(family))
    w = {}
    stage_blocks = {str(stage + 2): n for stage, n in enumerate(BLOCKS[backbone])}
    for conv, bn, kh, kw, cin, cout, _ in backbone_layers(backbone):
        w[conv + '/kernel'] = _normal(seed, conv, (kh, kw, cin, cout), np.sqrt(2.0 / (kh * kw * cin)))
        r = _rng(seed, bn)
        gamma = (1.0 + 0.1 * r.standard_normal(cout)).astype(np.float32)
        if bn.endswith('branch2c'):
            # keeps the residual stream O(1): 0.3 per block, less in the 23- / 36-block stages of ResNet-101 / -152
            n_blocks = stage_blocks[bn[2]]
            gamma *= np.float32(0.3 * min(1.0, np.sqrt(6.0 / n_blocks)))
        if bn.endswith('branch1'):
            gamma *= np.float32(0.7)
        w[bn + '/gamma'] = np.abs(gamma) + np.float32(0.05)
        w[bn + '/beta'] = (0.05 * r.standard_normal(cout)).astype(np.float32)
        w[bn + '/moving_mean'] = (0.05 * r.standard_normal(cout)).astype(np.float32)
        w[bn + '/moving_variance'] = (1.0 + 0.2 * r.random(cout)).astype(np.float32)
    # the image arrives un-normalised (BGR - mean, +-128): bring conv1 to unit scale
    w['conv1/kernel'] *= np.float32(1.0 / 64.0)
    for name, k, cin, cout, _ in fpn_layers():
        w[name + '/kernel'] = _normal(seed, name, (k, k, cin, cout), np.sqrt(1.0 / (k * k * cin)))
        w[name + '/bias'] = _normal(seed, name + '/bias', (cout,), 0.01)
    for name, cin, cout, kind in head_layers():
        if kind == 'tower':
            w[name + '/kernel'] = _normal(seed, name, (3, 3, cin, cout), np.sqrt(2.0 / (9 * cin)))
            w[name + '/bias'] = np.zeros((cout,), np.float32)
        else:
            gain = SYN_CLS_OUT_GAIN if 'classification' in name else (SYN_DIM_OUT_GAIN if 'dim' in name else SYN_REG_OUT_GAIN)
            gain *= SYN_BACKBONE_OUT_SCALE[backbone]
            w[name + '/kernel'] = _normal(seed, name, (3, 3, cin, cout), gain * np.sqrt(1.0 / (9 * cin)))
            w[name + '/bias'] = np.zeros((cout,), np.float32)
    # initializers.PriorProbability(0.01): bias = -log((1 - p) / p)   (initializers.py:23-39)
    w['pyramid_classification/bias'][:] = np.float32(-np.log((1.0 - 0.01) / 0.01))
    if family == 'trained':
        trained_like(w, backbone, seed)
    return w


TRAINED_STAGE_GAIN = (1.0, 4.0, 16.0, 64.0)       # scale of the residual stream in res2 .. res5 of the 'trained' family
TRAINED_CHANNEL_SPREAD = (1e-2, 1e1)              # per-channel scale of the inner bottleneck maps, log-uniform
TRAINED_DEAD_FRACTION = 0.02                      # channels of the inner maps that never fire


def trained_like(w, backbone, seed):
    """ Re-parameterise a synthetic draw IN PLACE so that its activation statistics look like a trained checkpoint's rather than like an
    initialisation (the reference's operating point is a trained KITTI model, README.md:75; none exists offline):
      * every channel of the two inner maps of a bottleneck (branch2a, branch2b outputs) gets its own scale, log-uniform over
        1e-2 .. 1e1 (BatchNormalization gamma and beta x s_c, the consuming kernel's input channel / s_c): three decades between the
        channels of one stored map -- what a narrow arithmetic type sees as small and large values side by side;
      * 2 % of those channels are dead (gamma = 0, beta < 0: zero after the ReLU);
      * the residual stream grows stage by stage (x 1, 4, 16, 64 for res2 .. res5: gamma and beta of branch2c / branch1 x G, the
        kernels that read the stream -- the next branch2a / branch1, C3 / C4 / C5_reduced, P6 -- / G).
    ReLU commutes with a positive scale, so the function the network computes is the base draw's up to rounding: the calibration of
    the synthetic head outputs (about a thousand anchors above the score threshold per frame) carries over. """
    r = _rng(seed, 'trained_like')
    lo, hi = np.log(TRAINED_CHANNEL_SPREAD[0]), np.log(TRAINED_CHANNEL_SPREAD[1])
    f32 = np.float32
    for stage, n_blocks in enumerate(BLOCKS[backbone]):
        G = f32(TRAINED_STAGE_GAIN[stage])
        for block in range(n_blocks):
            nm = block_name(backbone, stage, block)
            g_in = f32(TRAINED_STAGE_GAIN[stage - 1]) if (block == 0 and stage > 0) else (f32(1.0) if block == 0 else G)
            for producer, consumer in (('2a', '2b'), ('2b', '2c')):
                bn, conv = 'bn{}_branch{}'.format(nm, producer), 'res{}_branch{}'.format(nm, consumer)
                c = w[bn + '/gamma'].shape[0]
                s = np.exp(r.uniform(lo, hi, c)).astype(f32)
                dead = r.random(c) < TRAINED_DEAD_FRACTION
                w[bn + '/gamma'] = np.where(dead, f32(0.0), w[bn + '/gamma'] * s).astype(f32)
                w[bn + '/beta'] = np.where(dead, f32(-0.05), w[bn + '/beta'] * s).astype(f32)
                w[conv + '/kernel'] = (w[conv + '/kernel'] / s[None, None, :, None]).astype(f32)
            readers = ['res{}_branch2a'.format(nm)] + (['res{}_branch1'.format(nm)] if block == 0 else [])
            for conv in readers:
                w[conv + '/kernel'] = (w[conv + '/kernel'] / g_in).astype(f32)
            for bn in ['bn{}_branch2c'.format(nm)] + (['bn{}_branch1'.format(nm)] if block == 0 else []):
                w[bn + '/gamma'] = (w[bn + '/gamma'] * G).astype(f32)
                w[bn + '/beta'] = (w[bn + '/beta'] * G).astype(f32)
    for conv, stage in (('C3_reduced', 1), ('C4_reduced', 2), ('C5_reduced', 3), ('P6', 3)):
        w[conv + '/kernel'] = (w[conv + '/kernel'] / f32(TRAINED_STAGE_GAIN[stage])).astype(f32)
    return w


def parse_synthetic(spec):
    """ 'synthetic', 'synthetic:<seed>', 'synthetic:<seed>:trained' (also with a trailing '.h5': bin/run_network.py strips three
    characters of the model path) -> (seed, family) """
    import re
    m = re.match(r'synthetic(?::(\d+))?(?::(he|trained))?', spec)
    return (int(m.group(1)) if m and m.group(1) else 1234), (m.group(2) if m and m.group(2) else 'he')


def expected_arrays(backbone):
    """ {array name: shape} of every array the graph of `backbone` + FPN + heads needs (Keras names and layouts) """
    exp = {}
    for conv, bn, kh, kw, cin, cout, _ in backbone_layers(backbone):
        exp[conv + '/kernel'] = (kh, kw, cin, cout)
        for part in ('gamma', 'beta', 'moving_mean', 'moving_variance'):
            exp['{}/{}'.format(bn, part)] = (cout,)
    for name, k, cin, cout, _ in fpn_layers():
        exp[name + '/kernel'], exp[name + '/bias'] = (k, k, cin, cout), (cout,)
    for name, cin, cout, _ in head_layers():
        exp[name + '/kernel'], exp[name + '/bias'] = (3, 3, cin, cout), (cout,)
    return exp


def validate_weights(weights, backbone):
    """ Every expected array present with the expected shape, else ONE ValueError that names what is wrong (a converted
    checkpoint with a missing or transposed layer otherwise surfaces as a KeyError deep inside the plan builder). """
    exp = expected_arrays(backbone)
    missing = sorted(k for k in exp if k not in weights)
    wrong = sorted('{}: {} (expected {})'.format(k, tuple(np.shape(weights[k])), exp[k])
                   for k in exp if k in weights and tuple(np.shape(weights[k])) != exp[k])
    if missing or wrong:
        raise ValueError('weights do not match {} + FPN + heads: {} arrays missing{}{}; {} with a wrong shape{}{}'.format(
            backbone, len(missing), ': ' if missing else '', ', '.join(missing[:8]) + (' ...' if len(missing) > 8 else ''),
            len(wrong), ': ' if wrong else '', '; '.join(wrong[:8]) + (' ...' if len(wrong) > 8 else '')))


def _keras_group(layer):
    """ the /model_weights group that holds `layer`'s variables in a file written by the reference: its own, or that of the
    sub-model it lives in (models/retinanet.py:30,78,128; Keras stores e.g.
    /model_weights/regression_submodel/pyramid_regression_0/kernel:0) """
    if layer.startswith('pyramid_regression_dim'):
        return 'regression_dim_submodel'
    if layer.startswith('pyramid_regression'):
        return 'regression_submodel'
    if layer.startswith('pyramid_classification'):
        return 'classification_submodel'
    return layer


def _array_key(dataset_path):
    """ '<...>/<scope>/<variable>:0' -> '<scope>/<variable>' (the Keras variable name without its ':0' output index) """
    parts = dataset_path.split('/')
    if len(parts) < 2:
        raise ValueError('dataset {} is not a Keras variable (<layer>/<variable>:0)'.format(dataset_path))
    return '{}/{}'.format(parts[-2], parts[-1].split(':')[0])


def _collect(path, named_arrays):
    out = {}
    for name, arr in named_arrays:
        key = _array_key(name)
        if key in out:          # nested sub-models must not overwrite each other silently
            raise ValueError('{}: array {} appears twice (second time at {})'.format(path, key, name))
        out[key] = np.asarray(arr, dtype=np.float32)
    return out


def load_keras_h5(path):
    """ A Keras HDF5 model (`model.save`, what the reference's bin/convert_model.py:50-53 writes and models/__init__.py:81
    reads) or weights-only file (`model.save_weights`) -> {'<layer>/<variable>': float32 array}.
    Walks the file the way keras/engine/saving.py does: the `layer_names` attribute of the weight group, then each layer's
    `weight_names`; files without those attributes are walked dataset by dataset.  h5py when importable, else libhdf5 through
    ctypes (models/hdf5.py). """
    try:
        import h5py
    except ImportError:
        h5py = None
    if h5py is not None:
        found = []
        with h5py.File(path, 'r') as f:
            root = f['model_weights'] if 'model_weights' in f else f
            if 'layer_names' in root.attrs:
                for layer in root.attrs['layer_names']:
                    layer = layer.decode() if isinstance(layer, bytes) else str(layer)
                    g = root[layer]
                    for wn in g.attrs['weight_names']:
                        wn = wn.decode() if isinstance(wn, bytes) else str(wn)
                        found.append((layer + '/' + wn, g[wn][()]))
            else:
                root.visititems(lambda name, obj: found.append((name, obj[()])) if isinstance(obj, h5py.Dataset) else None)
        return _collect(path, found)
    from . import hdf5
    found = []
    with hdf5.File(path) as f:
        root = '/model_weights' if f.exists('/model_weights') else '/'
        layers = f.attr(root, 'layer_names')
        if layers is not None:
            for layer in layers:
                layer = layer.decode()
                group = root.rstrip('/') + '/' + layer
                names = f.attr(group, 'weight_names')
                if names is None:
                    raise ValueError('{}: layer group {} has no weight_names attribute'.format(path, group))
                for wn in names:
                    found.append((layer + '/' + wn.decode(), f.read(group + '/' + wn.decode())))
        else:
            found = sorted(f.datasets(root).items())
    return _collect(path, found)


def save_keras_h5(path, weights, backbone=None):
    """ Writes `weights` in the layout of a Keras `model.save` file (see models/hdf5.py): /model_weights/<layer>/<scope>/<var>:0
    with the layer_names / weight_names attributes, head layers below their sub-model groups as the reference nests them.
    Carries the weights only (no model_config): it is what load_keras_h5 / load_model read back, and a fixture in the
    reference's format -- not a file `keras.models.load_model` could rebuild a graph from. """
    from . import hdf5
    order = ('kernel', 'bias', 'gamma', 'beta', 'moving_mean', 'moving_variance')      # Keras variable order within a layer
    groups = {}
    for key in weights:
        layer, var = key.rsplit('/', 1)
        groups.setdefault(_keras_group(layer), []).append((layer, var))
    with hdf5.File(path, 'w') as f:
        f.create_group('/model_weights')
        for at in ('/', '/model_weights'):
            f.write_attr(at, 'keras_version', b'2.2.0')
            f.write_attr(at, 'backend', b'tensorflow')
        f.write_attr('/model_weights', 'layer_names', [g.encode() for g in groups])
        for group, members in groups.items():
            members.sort(key=lambda lv: (lv[0], order.index(lv[1]) if lv[1] in order else len(order)))
            f.create_group('/model_weights/' + group)
            f.write_attr('/model_weights/' + group, 'weight_names', ['{}/{}:0'.format(l, v).encode() for l, v in members])
            for layer, var in members:
                f.write_dataset('/model_weights/{}/{}/{}:0'.format(group, layer, var), weights['{}/{}'.format(layer, var)])


def load_weights(path):
    """ '.npz' written by save_weights, or a Keras '.h5' / '.hdf5' model or weight file (load_keras_h5) """
    if path.endswith('.npz'):
        with np.load(path) as z:
            return {k: np.asarray(z[k], dtype=np.float32) for k in z.files}
    if path.endswith('.h5') or path.endswith('.hdf5'):
        return load_keras_h5(path)
    raise ValueError('unknown weight file type: {}'.format(path))


def save_weights(path, weights):
    """ '.npz' (NumPy) or '.h5' / '.hdf5' (Keras layout, save_keras_h5) by extension """
    if path.endswith('.h5') or path.endswith('.hdf5'):
        return save_keras_h5(path, weights)
    np.savez(path, **weights)


def folded_conv(weights, conv, bn=None):
    """ (kernel HWIO f32, bias f32) of a convolution with its frozen BatchNormalization folded in:
        y = gamma * (conv(x) - mean) / sqrt(var + eps) + beta  =  conv(x, k * s) + (beta - mean * s) """
    k = np.asarray(weights[conv + '/kernel'], dtype=np.float32)
    if bn is None:
        return k, np.asarray(weights[conv + '/bias'], dtype=np.float32)
    s = weights[bn + '/gamma'].astype(np.float64) / np.sqrt(weights[bn + '/moving_variance'].astype(np.float64) + BN_EPSILON)
    bias = weights[bn + '/beta'].astype(np.float64) - weights[bn + '/moving_mean'].astype(np.float64) * s
    return (k.astype(np.float64) * s[None, None, None, :]).astype(np.float32), bias.astype(np.float32)


def fused_regression_outputs(weights):
    """ The five regression output convolutions (retinanet.py:112-120) as one C_out = 144 layer,
    channel order [op1: 4A | op2: 2A | op3: 2A | op4: 2A | op5: 2A]. """
    names = ['pyramid_regression_op{}'.format(k) for k in (1, 2, 3, 4, 5)]
    kernel = np.concatenate([weights[n + '/kernel'] for n in names], axis=3)
    bias = np.concatenate([weights[n + '/bias'] for n in names], axis=0)
    return kernel.astype(np.float32), bias.astype(np.float32)


def fused_tower_inputs(weights):
    """ The first layer of the three head towers (pyramid_regression_0, pyramid_classification_0,
    pyramid_regression_dim_0; retinanet.py:100-107,52-60,151-158) read the same pyramid feature: one
    C_out = 512 + 256 + 128 = 896 layer, channel order [regression | classification | dimension]. """
    names = ['pyramid_regression_0', 'pyramid_classification_0', 'pyramid_regression_dim_0']
    kernel = np.concatenate([weights[n + '/kernel'] for n in names], axis=3)
    bias = np.concatenate([weights[n + '/bias'] for n in names], axis=0)
    return kernel.astype(np.float32), bias.astype(np.float32)
