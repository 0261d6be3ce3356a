""" Which tensors differ between two models with different tile choices / batch splits?  (debugging aid) """
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'ground-plane-polling_amd'))
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from keras_retinanet_3D import models  # noqa: E402
import sharded_worker  # noqa: E402

dtype = sys.argv[1] if len(sys.argv) > 1 else 'bf16'
inputs = list(sharded_worker.global_inputs(4, 402, 1333))
names = ['boxes', 'dims', 'scores', 'labels', 'orient', 'keypoints', 'keyplanes', 'residuals']


def run(model, inp):
    outs = model.predict_on_batch(inp)
    B = inp[0].shape[0]
    plan = model.plan_for(B, 402, 1333, 1000, True)
    extra = {'cls': plan.cls_logits.cpu().numpy(), 'reg': plan.regression.cpu().numpy(), 'dim': plan.regression_dim.cpu().numpy(),
             'best': plan.best_index.cpu().numpy(), 'anchor': plan.anchor_index.cpu().numpy()}
    feats = {k: v.dense().float().cpu().numpy() for k, v in plan.features.items() if k in ('C3', 'C4', 'C5', 'P3', 'P4', 'P5', 'P6', 'P7')}
    return outs, extra, feats, plan


def diff(tag, a, b):
    bad = False
    for k in range(8):
        if a[0][k].tobytes() != b[0][k].tobytes():
            d = np.abs(a[0][k].astype(np.float64) - b[0][k].astype(np.float64))
            print(tag, 'OUTPUT', names[k], 'differs: n =', int((d > 0).sum()), 'max', d.max(), 'first at', np.argwhere(d > 0)[0])
            bad = True
    for k in a[1]:
        if a[1][k].tobytes() != b[1][k].tobytes():
            d = np.abs(a[1][k].astype(np.float64) - b[1][k].astype(np.float64))
            print(tag, 'HEAD', k, 'differs: n =', int((d > 0).sum()), 'max', d.max())
            bad = True
    for k in a[2]:
        if a[2][k].tobytes() != b[2][k].tobytes():
            d = np.abs(a[2][k].astype(np.float64) - b[2][k].astype(np.float64))
            print(tag, 'FEATURE', k, 'differs: n =', int((d > 0).sum()), 'max', d.max())
            bad = True
    print(tag, 'DIFFERENT' if bad else 'identical')


mA = models.load_model('synthetic:1234', dtype=dtype)
rA = run(mA, inputs)
print('tiles A:', {k: v[0] for k, v in rA[3].tuning.items()})
os.environ['GPP_AUTOTUNE'] = '0'
mB = models.load_model('synthetic:1234', dtype=dtype)
rB = run(mB, inputs)
diff('autotuned vs heuristic tiles (B=4):', rA, rB)
for rep in range(3):
    diff('autotuned again %d:' % rep, rA, run(mA, inputs))
del os.environ['GPP_AUTOTUNE']
mC = models.load_model('synthetic:1234', dtype=dtype)
rC = run(mC, inputs)
print('tiles C:', {k: v[0] for k, v in rC[3].tuning.items() if rA[3].tuning[k][0] != v[0]})
diff('second autotuned model (B=4):', rA, rC)
# shards
for lo in (0, 2):
    sub = [a[lo:lo + 2] for a in inputs]
    rS = run(mA, sub)
    whole = ([o[lo:lo + 2] for o in rA[0]], {k: v[lo:lo + 2] for k, v in rA[1].items()}, {k: v[lo:lo + 2] for k, v in rA[2].items()})
    diff('shard %d..%d of model A vs its slice of the B=4 run:' % (lo, lo + 2), whole, rS)
