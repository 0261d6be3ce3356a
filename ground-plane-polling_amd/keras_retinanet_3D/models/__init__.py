"""
Model loading surface of the reference, kept name for name for the inference path
(/root/reference/keras_retinanet_3D/models/__init__.py:9-88):

    backbone(name)                       -> Backbone object (validate(), retinanet())
    load_model(filepath, backbone_name)  -> object with predict_on_batch([images, P_inv, planes])

`filepath` may be
    'synthetic:<seed>'   seeded random weights of the exact architecture (no trained weights ship
                         with the reference, README.md:75); 'synthetic:<seed>:trained' = the same draw with the activation statistics
                         of a trained checkpoint (per-channel scales three decades apart, dead channels, a residual stream that
                         grows stage by stage: models/weights.trained_like)
    '<file>.npz'         weights saved by models.weights.save_weights (Keras layer names)
    '<file>.h5'          a Keras model (`model.save`) or weight (`model.save_weights`) file, the format the reference reads and
                         writes (bin/convert_model.py:50-53): h5py when importable, else libhdf5 through ctypes (models/hdf5.py)
"""


class Backbone(object):
    """ This class stores additional information on backbones (reference models/__init__.py:9-39). """

    def __init__(self, backbone):
        self.backbone = backbone
        self.custom_objects = {}      # Keras deserialisation table of the reference; nothing to register here
        self.validate()

    def retinanet(self, *args, **kwargs):
        raise NotImplementedError('retinanet method not implemented.')

    def download_imagenet(self):
        raise NotImplementedError('download_imagenet method not implemented.')

    def validate(self):
        raise NotImplementedError('validate method not implemented.')


def backbone(backbone_name):
    """ Returns a backbone object for the given backbone (reference models/__init__.py:42-56). """
    if 'resnet' in backbone_name:
        from .resnet import ResNetBackbone as b
    else:
        raise NotImplementedError('Backbone class for  \'{}\' not implemented.'.format(backbone_name))
    return b(backbone_name)


def load_model(filepath, backbone_name='resnet50', convert=False, nms=True, class_specific_filter=True,
               orientation_specific_filter=False, dtype=None, on_range_event=None, plan=None):
    """ Loads a RetinaNet-3D inference model (reference models/__init__.py:59-88).

    `convert` is accepted for signature compatibility: every model this function returns already
    contains the decode / NMS / ground-plane-polling stages (`retinanet_bbox`, retinanet.py:359-422).
    `dtype` (not in the reference; None = the environment's GPP_DTYPE, else 'f16x3'):
        'f16x3' (default)  float32-sized storage, every float32 product as three IEEE-half matrix products: the fastest type whose
                           detections, plane indices and 3-D corners stay within BASELINE's tolerance of the float32 path
        'f32'              float32 storage and operands (the reference's floatx, /root/reference/keras_retinanet_3D/utils/image.py:47)
        'bf16x3'           three bf16 products per float32 product (~2^-16): same detections and planes, corners off by millimetres
        'f16' | 'bf16'     16-bit storage and MFMA operands, float32 accumulation: 2.4x faster, a few percent of the detections differ
    `on_range_event` (not in the reference; dtype='f16x3' only; None = the environment's GPP_ON_RANGE_EVENT, else 'f32'): what a call does
    when one of its activations left the IEEE-half range (beyond +-65504: clamped, counted by the kernels' epilogues) --
        'f32' (default)    the call is run again on a float32 twin of the model and THAT result is returned (the reference's answer, slower)
        'raise'            GppError;      'ignore'   the clamped result is returned (model.x3_range_events() still counts)
    `plan` (not in the reference; None = the environment's GPP_PLAN, else 'throughput'): how the layers are launched --
        'throughput'       (default) the plan every batched caller wants: split-K only where a layer's grid is tiny at ANY batch
        'latency'          for callers that time ONE image per call, as the reference does (bin/run_network.py:108-111): the deep-K layers whose
                           batch-1 grid leaves most of the 256 CUs idle (res4 / res5 bottlenecks, P4 ... P7, the lateral 1x1s) split their K
                           loop over more workgroups.  The split is a rule of (layer, plan) -- never of the batch, the tile or a timing -- so a
                           model gives byte-identical results at every batch size and on every rank WITHIN its plan mode; between the two modes
                           results differ by float32 summation order only (both inside the parity bars, tests/test_latency_plan_gpu.py)
    """
    import os
    if dtype is None:
        dtype = os.environ.get('GPP_DTYPE', 'f16x3')
    from . import weights as W
    from .retinanet import RetinaNet3D
    b = backbone(backbone_name)
    name = b.backbone.split('_')[0]
    if isinstance(filepath, dict):
        w = filepath
    elif isinstance(filepath, str) and filepath.startswith('synthetic'):
        seed, family = W.parse_synthetic(filepath)         # 'synthetic:7', 'synthetic:7:trained', also 'synthetic:7.h5' (run_network strips 3 chars)
        w = W.synthetic_weights(name, seed, family)
    else:
        w = W.load_weights(filepath)
    model = RetinaNet3D(w, backbone_name=name, dtype=dtype, nms=nms, class_specific_filter=class_specific_filter,
                        orientation_specific_filter=orientation_specific_filter, on_range_event=on_range_event, plan=plan)
    if convert:
        model.summary()
    return model
