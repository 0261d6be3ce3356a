"""
Times the host-fed streaming loop (utils.pipeline.FramePipeline) in its variants against the HBM-resident step rate:
    python tools/pipe_variants.py [batches]
Prints images/s for: resident plan runs, predict_on_frames (synchronous), and the pipeline with depth 2/3/4, with and
without HIP-graph replay of the plan, with and without the pinned download buffer.
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'ground-plane-polling_amd'))
import numpy as np   # noqa: E402
import torch   # noqa: E402

from keras_retinanet_3D import models   # noqa: E402
from keras_retinanet_3D.utils import synthetic   # noqa: E402
from keras_retinanet_3D.utils.pipeline import FramePipeline   # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
B = 8
model = models.load_model('synthetic:1234', dtype='bf16')
planes = synthetic.load_plane_database('1k').astype(np.float32)
frames = (np.random.default_rng(5).integers(0, 2, size=(B, 375, 1242, 3)) * 255).astype(np.uint8)
_, P_inv = synthetic.synthetic_calibration(1333.0 / 1242.0)
P = np.tile(P_inv[None].astype(np.float32), (B, 1, 1))
pl = np.tile(planes[None], (B, 1, 1))
for _ in range(3):
    model.predict_on_frames(frames, P, pl)
plan = model.plan_for(B, 402, 1333, 1000, True)
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(n):
    model.run_plan(plan)
torch.cuda.synchronize()
print('resident            %.1f images/s' % (B * n / (time.perf_counter() - t)))
t = time.perf_counter()
for _ in range(n):
    model.predict_on_frames(frames, P, pl)
torch.cuda.synchronize()
print('predict_on_frames   %.1f images/s' % (B * n / (time.perf_counter() - t)))
for graph in (False,):
    for pinned in (True, False):
        for depth in (3, 4, 5):
            pipe = FramePipeline(model, depth=depth, graph=graph, pinned=pinned)
            list(pipe.run(iter([(frames, P, pl)] * 4)))
            torch.cuda.synchronize()
            best = 0.0
            for rep in range(2):
                t = time.perf_counter()
                for _ in pipe.run(iter([(frames, P, pl)] * n)):
                    pass
                best = max(best, B * n / (time.perf_counter() - t))
            print('pipeline depth %d graph %d pinned %d   %.1f images/s' % (depth, graph, pinned, best))
    plan.graph = None
