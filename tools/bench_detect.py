""" Micro-benchmark of gpp_detect_f32 (candidates + NMS kernels) on synthetic head tensors. """
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'ground-plane-polling_amd'))
import numpy as np
import torch
from keras_retinanet_3D.layers.filter_detections import FilterDetections
from keras_retinanet_3D.utils import anchors as A

dev = torch.device('cuda')
anchors = torch.as_tensor(A.anchors_for_image((402, 1333))).to(dev)
n = anchors.shape[0]
B = 8
g = torch.Generator(device='cpu').manual_seed(0)
for name, mean, std, thr, md in (('K~1000/img', -4.6, 0.52, 0.05, 100), ('K~1000/img max_det=1', -4.6, 0.52, 0.05, 1), ('K~1000/img max_det=10', -4.6, 0.52, 0.05, 10), ('no candidates (thr 0.9)', -4.6, 0.52, 0.9, 100), ('K~40000/img', -4.0, 0.62, 0.05, 100)):
    logits = (torch.randn((B, n, 8), generator=g) * std + mean).to(dev)
    reg = torch.randn((B, n // 12, 144), generator=g).to(dev)
    dim = torch.randn((B, n, 3), generator=g).to(dev)
    op = FilterDetections(B, n, dev, fused_regression=True, score_threshold=thr, max_detections=md)
    for _ in range(3):
        op(logits, reg, dim, anchors)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        op(logits, reg, dim, anchors)
    e1.record(); torch.cuda.synchronize()
    print('%-26s %.1f us per call (candidates + nms), counts %s' % (name, e0.elapsed_time(e1) / 20 * 1e3, op.counts[:3].tolist()))
