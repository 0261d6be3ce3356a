"""
Child process of tests/test_zz_sharded_gpu.py::test_rccl_branch_of_the_gather: a world of ONE rank on the RCCL ('nccl') backend, i.e.
the code path the driver's multi-GPU run takes (utils.distributed.gather_detections -> all_gather_into_tensor of the packed
(B, 100, 35) detections on the device, synchronous and asynchronous form), on the one GPU of the test box.
    python rccl_worker.py <port> <batch> <H> <W> <dtype> <out.npy>
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, 'ground-plane-polling_amd'), os.path.join(ROOT, 'tests'), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)


def main():
    port, batch, h, w = (int(v) for v in sys.argv[1:5])
    dtype, out_path = sys.argv[5], sys.argv[6]
    import numpy as np
    import torch
    import torch.distributed as dist
    import sharded_worker
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK='0', WORLD_SIZE='1', HSA_ENABLE_IPC_MODE_LEGACY='0')
    torch.cuda.set_device(0)
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
    from keras_retinanet_3D import models
    from keras_retinanet_3D.utils import distributed as D
    model = models.load_model('synthetic:1234', backbone_name='resnet50', dtype=dtype)
    inputs = list(sharded_worker.global_inputs(batch, h, w))
    outs = D.ShardedModel(model).predict_on_batch(inputs)                      # synchronous all_gather_into_tensor
    packed = np.concatenate([np.asarray(o, np.float32).reshape(batch, 100, -1) for o in outs], axis=2)
    plan = model.plan_for(batch, h, w, 1000, True)
    local = D.pack_outputs(model.outputs(plan))
    gathered, work = D.gather_detections(local, async_op=True)                 # the form bench.py overlaps with the next step
    work.wait()
    torch.cuda.synchronize()
    assert dist.get_backend() == 'nccl' and gathered.is_cuda and tuple(gathered.shape) == (batch, 100, D.PACK_WIDTH)
    np.save(out_path, np.stack([packed, gathered.cpu().numpy()]))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
