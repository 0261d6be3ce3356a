"""
keras_retinanet_3D -- MI355X-native implementation of the Ground-Plane-Polling inference path.

Same import surface as the reference package for that path
(`keras_retinanet_3D.models.load_model(...).predict_on_batch([images, P_inv, planes])`,
`keras_retinanet_3D.utils.gpp_utils`, `keras_retinanet_3D.bin.run_network`); the compute runs
in hand-written HIP kernels (../csrc) behind the C ABI of include/gpp.h.
"""

__version__ = '0.1.0'
