""" Compare intermediate feature maps of the device plan with the CPU oracle (run on the GPU box). """
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'ground-plane-polling_amd')); sys.path.insert(0, ROOT)
from oracle import net_torch
from keras_retinanet_3D import models
from keras_retinanet_3D.models import weights as W
from keras_retinanet_3D.utils import synthetic
B, H, Wd = 2, 96, 160
rng = np.random.default_rng(96)
img = rng.integers(0, 256, size=(B, H, Wd, 3)).astype(np.float32) - np.array([103.939, 116.779, 123.68], np.float32)
w = W.synthetic_weights('resnet50', 1234)
m = models.load_model(w, backbone_name='resnet50', dtype='bf16')
_, P_inv = synthetic.synthetic_calibration()
plan = m.stage_inputs([img, np.tile(P_inv[None], (B, 1, 1)), synthetic.load_plane_database('10')])
m.run_plan(plan)
q = net_torch.forward(w, img, 'resnet50', storage='bf16', keep_features=True)
f = net_torch.forward(w, img, 'resnet50', storage=None, keep_features=True)
for k in ['C2', 'C3', 'C4', 'C5', 'P3', 'P4', 'P5', 'P6', 'P7']:
    g = plan.features[k].dense().float().cpu().numpy()
    e = np.abs(g - q[k]); ef = np.abs(g - f[k]); eqf = np.abs(q[k] - f[k])
    print('{:3s} shape {} rms {:.3f} | gpu-q max {:.4f} med {:.5f} frac>0 {:.3f} | gpu-f32 max {:.4f} med {:.5f} | q-f32 max {:.4f} med {:.5f}'.format(
        k, g.shape, np.sqrt((q[k]**2).mean()), e.max(), np.median(e), (e > 0).mean(), ef.max(), np.median(ef), eqf.max(), np.median(eqf)))
