""" Repeats the two-process shard scenario of tests/test_zz_sharded_gpu.py N times against one single-process reference and says
what differs if anything does (debugging aid for a once-seen mismatch):  python tools/shard_stress.py [N] [dtype] """
import os
import socket
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'ground-plane-polling_amd'))
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np  # noqa: E402
import sharded_worker  # noqa: E402
from keras_retinanet_3D import models  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 6
dtype = sys.argv[2] if len(sys.argv) > 2 else 'bf16'
batch, h, w = 4, 402, 1333
model = models.load_model('synthetic:1234', backbone_name='resnet50', dtype=dtype)
outs = model.predict_on_batch(list(sharded_worker.global_inputs(batch, h, w)))
single = np.concatenate([np.asarray(o, np.float32).reshape(batch, 100, -1) for o in outs], axis=2)
bad = 0
for it in range(n):
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    tmp = tempfile.mkdtemp()
    out_path = os.path.join(tmp, 'g.npy')
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0', GPP_SHARD_DEBUG_DIR=tmp)
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, 'tests', 'sharded_worker.py'), str(r), '2', str(port), str(batch), str(h), str(w),
                               dtype, out_path], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(2)]
    for p in procs:
        p.communicate(timeout=600)
    g = np.load(out_path)
    if g.tobytes() == single.tobytes():
        print('iteration %d: identical' % it, flush=True)
        continue
    bad += 1
    d = np.abs(g.astype(np.float64) - single.astype(np.float64))
    rows = np.argwhere(d.max(axis=2) > 0)
    print('iteration %d: MISMATCH rows %s cols %s max %g' % (it, rows[:6].tolist(), sorted(set(np.argwhere(d > 0)[:, 2].tolist())), d.max()), flush=True)
    for r in range(2):
        z = np.load(os.path.join(tmp, 'rank%d.npz' % r))
        plan = model.plan_for(batch, h, w, 1000, True)
        lo = 2 * r
        print('   rank %d: planes equal %s canon equal %s P_inv equal %s boxes equal %s best equal %s' % (
            r, np.array_equal(z['planes'], plan.planes.cpu().numpy()[lo:lo + 2]),
            np.array_equal(z['canon'], plan.poll_ws.cpu().numpy().view(np.float32).reshape(4, -1)[lo:lo + 2].reshape(-1)),
            np.array_equal(z['P_inv'], plan.P_inv.cpu().numpy()[lo:lo + 2]), np.array_equal(z['boxes'], plan.boxes.cpu().numpy()[lo:lo + 2]),
            np.array_equal(z['best'], plan.best_index.cpu().numpy()[lo:lo + 2])), flush=True)
    again = model.predict_on_batch(list(sharded_worker.global_inputs(batch, h, w)))
    second = np.concatenate([np.asarray(o, np.float32).reshape(batch, 100, -1) for o in again], axis=2)
    print('   single-process run repeated: equals its first result %s, equals the gathered one %s' % (
        second.tobytes() == single.tobytes(), second.tobytes() == g.tobytes()), flush=True)
print('%d of %d iterations mismatched' % (bad, n))
