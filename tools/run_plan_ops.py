""" Run named launches of the B = 8 f16x3 plan in isolation (for PMC passes: tools/pmc_layers.sh): each op 8 times back to back, then 4 times
cold (600 MB rewritten in between).  Prints layer -> tile and the hot / cold time.
    python tools/run_plan_ops.py <dtype> <op name> [<op name> ...] """
import os
import sys
os.environ.setdefault('GPP_HALF_LANES', '')           # one launch per layer (whole batch)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'ground-plane-polling_amd'))
import torch  # noqa: E402
from keras_retinanet_3D import models  # noqa: E402

dtype, names = sys.argv[1], sys.argv[2:]
model = models.load_model('synthetic:1234', backbone_name='resnet50', dtype=dtype)
plan = model.plan_for(8, 402, 1333, 1000, True)
model.run_plan(plan)
torch.cuda.synchronize()
flush = torch.empty((600 << 20,), dtype=torch.uint8, device='cuda')
index = {op[3]: i for i, op in enumerate(plan.ops)}
for name in names:
    i = index[name]
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    model.run_op(plan, i)
    e0.record()
    for _ in range(8):
        model.run_op(plan, i)
    e1.record()
    torch.cuda.synchronize()
    hot = e0.elapsed_time(e1) * 1e3 / 8
    cold = []
    for _ in range(4):
        flush.fill_(1)
        e0.record()
        model.run_op(plan, i)
        e1.record()
        torch.cuda.synchronize()
        cold.append(e0.elapsed_time(e1) * 1e3)
    print('{:28s} tile {:8d}  hot {:7.1f} us  cold {:7.1f} us'.format(name, plan.tuning.get(name, (0, 0))[0], hot, sorted(cold)[1]))
