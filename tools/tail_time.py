import sys, os
sys.path.insert(0, 'ground-plane-polling_amd')
import torch
from keras_retinanet_3D import models
for ft in ('0', '64'):
    os.environ['GPP_FUSE_TAIL'] = ft
    m = models.load_model('synthetic:1234', dtype='f16x3')
    p = m.plan_for(8, 402, 1333, 1000, True)
    print('GPP_FUSE_TAIL=' + ft, {k: v for k, v in p.tuning.items() if k.startswith('res2')})
    flush = torch.empty((600 << 20,), dtype=torch.uint8, device='cuda')
    names = [n for _, _, _, n, _ in p.ops]
    for i, n in enumerate(names):
        if not n.startswith('res2b'):
            continue
        ts = []
        for _ in range(5):
            flush.fill_(1)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); m.run_op(p, i); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
        print('   cold', n, '%.1f us' % sorted(ts)[2])
    del m, p
