import sys, time, os
sys.path.insert(0, 'ground-plane-polling_amd'); sys.path.insert(0, '.')
import numpy as np, torch
import bench
from keras_retinanet_3D import models
from keras_retinanet_3D.utils import synthetic
dtype = sys.argv[1] if len(sys.argv) > 1 else 'f16x3'
model = models.load_model('synthetic:1234', backbone_name='resnet50', dtype=dtype)
planes = synthetic.load_plane_database('1k').astype(np.float32)
_, P_inv = synthetic.synthetic_calibration()
B = 8
images = torch.as_tensor(bench.synthetic_batch(B, 0)).cuda()
P = torch.as_tensor(np.tile(P_inv[None].astype(np.float32), (B, 1, 1))).cuda()
pl = torch.as_tensor(np.tile(planes[None], (B, 1, 1))).cuda()
plan = model.stage_inputs([images, P, pl])
def timeit(n=40):
    for _ in range(5): model.run_plan(plan)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): model.run_plan(plan)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
a = timeit(); ref = [t.clone() for t in model.outputs(plan)]
model.capture(plan)
b = timeit(); same = all(torch.equal(x, y) for x, y in zip(ref, model.outputs(plan)))
plan.graph = None
c = timeit()
print(dtype, 'eager %.3f ms  graph %.3f ms  eager again %.3f ms  same outputs %s' % (a, b, c, same))
