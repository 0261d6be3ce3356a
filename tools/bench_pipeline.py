""" Host-fed throughput: FramePipeline (overlapped uploads) vs synchronous predict_on_frames. """
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'ground-plane-polling_amd'))
import numpy as np, torch
from keras_retinanet_3D import models
from keras_retinanet_3D.utils import synthetic
from keras_retinanet_3D.utils.pipeline import FramePipeline
m = models.load_model('synthetic:1234', backbone_name='resnet50')
B = 8
frames = np.stack([synthetic.synthetic_image(seed=i) for i in range(B)])
planes = np.tile(synthetic.load_plane_database('1k').astype(np.float32)[None], (B, 1, 1))
_, P = synthetic.synthetic_calibration(1333/1242); P = np.tile(P[None].astype(np.float32), (B,1,1))
pipe = FramePipeline(m, 2)
list(pipe.run(iter([(frames, P, planes)]*3))); torch.cuda.synchronize()
t0 = time.perf_counter(); stamps = []
for k, out in enumerate(pipe.run(iter([(frames, P, planes)] * 30))):
    stamps.append(time.perf_counter() - t0)
print('pipelined loop: total %.1f ms for 30 batches (%.0f img/s); per-yield ms: %s' % (stamps[-1]*1e3, 240/stamps[-1], ' '.join('%.1f' % ((b-a)*1e3) for a, b in zip([0]+stamps[:-1], stamps))))
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(30): m.predict_on_frames(frames, P, planes)
print('synchronous loop: %.0f img/s' % (240/(time.perf_counter()-t0)))
