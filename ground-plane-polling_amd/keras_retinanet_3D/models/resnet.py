"""
ResNet backbones (reference models/resnet.py:24-114).  The arithmetic of the backbone is the
third-party keras_resnet package in the reference; here it is part of the device plan built by
models/retinanet.py from the layer inventory in models/weights.py.
"""

from . import Backbone

ALLOWED_BACKBONES = ['resnet50', 'resnet101', 'resnet152']


class ResNetBackbone(Backbone):
    """ Describes backbone information and provides utility functions. """

    def retinanet(self, *args, **kwargs):
        """ Returns a retinanet model using the correct backbone. """
        return resnet_retinanet(*args, backbone=self.backbone, **kwargs)

    def validate(self):
        """ Checks whether the backbone string is correct (reference models/resnet.py:61-68). """
        name = self.backbone.split('_')[0]
        if name not in ALLOWED_BACKBONES:
            raise ValueError('Backbone (\'{}\') not in allowed backbones ({}).'.format(name, ALLOWED_BACKBONES))


def resnet_retinanet(num_classes=1, backbone='resnet50', weights='synthetic:1234', **kwargs):
    """ Constructs a RetinaNet-3D inference model using a resnet backbone. """
    if num_classes != 1:
        raise NotImplementedError('one object class (the reference\'s only trained configuration)')
    from . import load_model
    return load_model(weights, backbone_name=backbone, **kwargs)


def resnet50_retinanet(num_classes=1, **kwargs):
    return resnet_retinanet(num_classes=num_classes, backbone='resnet50', **kwargs)


def resnet101_retinanet(num_classes=1, **kwargs):
    return resnet_retinanet(num_classes=num_classes, backbone='resnet101', **kwargs)


def resnet152_retinanet(num_classes=1, **kwargs):
    return resnet_retinanet(num_classes=num_classes, backbone='resnet152', **kwargs)
