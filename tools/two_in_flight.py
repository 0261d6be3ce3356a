""" Experiment: two batches in flight.  Step k runs on stream k % 2 with its own plan (own activation buffers), so the backbone of one
batch (bound by tile fills and latency) overlaps the head towers of the other (bound by the matrix pipe and, on real data, by the clock under
matrix load).  Compared with the same number of steps on one stream.      python tools/two_in_flight.py [dtype] [steps] """
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'ground-plane-polling_amd'))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
from keras_retinanet_3D import models  # noqa: E402
from keras_retinanet_3D.utils import synthetic  # noqa: E402
import bench  # noqa: E402

dtype = sys.argv[1] if len(sys.argv) > 1 else 'f16x3'
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
B = 8
planes = synthetic.load_plane_database('1k').astype(np.float32)
_, P_inv = synthetic.synthetic_calibration()
ms, plans = [], []
for j in range(2):
    m = models.load_model('synthetic:1234', backbone_name='resnet50', dtype=dtype)
    images = torch.as_tensor(bench.synthetic_batch(B, 100 * j)).cuda()
    P = torch.as_tensor(np.tile(P_inv[None].astype(np.float32), (B, 1, 1))).cuda()
    pl = torch.as_tensor(np.tile(planes[None], (B, 1, 1))).cuda()
    plans.append(m.stage_inputs([images, P, pl]))
    ms.append(m)
torch.cuda.synchronize()
streams = [torch.cuda.Stream(), torch.cuda.Stream()]


def run(two):
    for rep in range(2):                                  # the first repetition warms up
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(steps):
            j = k % 2
            with torch.cuda.stream(streams[j if two else 0]):
                ms[j].run_plan(plans[j])
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    return B * steps / dt, dt / steps * 1e3


ref = [[t.clone() for t in ms[j].outputs(plans[j])] for j in range(2)]
for rep in range(3):
    a = run(False)
    if rep == 0:
        ref = [[t.clone() for t in ms[j].outputs(plans[j])] for j in range(2)]
    b = run(True)
    same = all(torch.equal(x.view(torch.uint8), y.view(torch.uint8)) for j in range(2) for x, y in zip(ref[j], ms[j].outputs(plans[j])))
    print('%s: one stream %.1f images/s (%.3f ms/step)   two batches in flight %.1f images/s (%.3f ms/step)   %+.1f %%   outputs identical: %s' %
          (dtype, a[0], a[1], b[0], b[1], (b[0] / a[0] - 1) * 100, same))
