"""
FitRoadPlanes: per-detection ground-plane polling on the device (HIP kernel csrc/poll.hip).
Same role and input order as the reference layer
/root/reference/keras_retinanet_3D/layers/fit_road_planes.py:142-186.
"""

from ..utils.gpp_utils import fit_road_planes


class FitRoadPlanes(object):
    def __init__(self, name='fit_road_planes'):
        self.name = name

    def call(self, inputs):
        """ inputs: [boxes, dimensions, orientations, P_inv, planes] (fit_road_planes.py:157-161) """
        boxes, dimensions, orientations, P_inv, planes = inputs
        return fit_road_planes(boxes, dimensions, orientations, P_inv, planes)

    __call__ = call

    def compute_output_shape(self, input_shape):
        b, d = input_shape[0][0], input_shape[0][1]
        return [(b, d, 4, 3), (b, d, 1, 4), (b, d)]
