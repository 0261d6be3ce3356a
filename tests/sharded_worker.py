"""
Child process of tests/test_zz_sharded_gpu.py: one rank of an image-sharded run of the REAL model.
    python sharded_worker.py <rank> <world> <port> <global_batch> <H> <W> <dtype> <out.npy>
Every rank builds its own model (its own block-tile tuning run), takes its contiguous shard of the seeded global batch,
and the ranks exchange the packed detections over a gloo group (both ranks share the one GPU of the test box; the
driver's 8-GPU run uses RCCL through the same utils.distributed code).  Rank 0 writes the gathered (B, 100, 35) tensor.
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, 'ground-plane-polling_amd'), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)


def global_inputs(batch, h, w):
    import numpy as np
    from keras_retinanet_3D.utils import synthetic
    rng = np.random.default_rng(99)
    img = rng.integers(0, 256, size=(batch, h, w, 3)).astype(np.float32) - np.array([103.939, 116.779, 123.68], np.float32)
    planes = synthetic.load_plane_database('1k').astype(np.float32)
    _, P_inv = synthetic.synthetic_calibration()
    return img, np.tile(P_inv[None].astype(np.float32), (batch, 1, 1)), np.tile(planes[None], (batch, 1, 1))


def main():
    rank, world, port, batch, h, w = (int(v) for v in sys.argv[1:7])
    dtype, out_path = sys.argv[7], sys.argv[8]
    import numpy as np
    import torch
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    os.environ['RANK'] = str(rank)
    torch.cuda.set_device(0)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from keras_retinanet_3D import models
    from keras_retinanet_3D.utils import distributed as D
    model = models.load_model('synthetic:1234', backbone_name='resnet50', dtype=dtype)
    sharded = D.ShardedModel(model)
    outs = sharded.predict_on_batch(list(global_inputs(batch, h, w)))
    if rank == 0:
        packed = np.concatenate([np.asarray(o, np.float32).reshape(batch, 100, -1) for o in outs], axis=2)
        np.save(out_path, packed)
    debug = os.environ.get('GPP_SHARD_DEBUG_DIR')
    if debug:                    # the polling stage's shared inputs and this rank's own outputs, as they are on the device
        lo, hi = D.shard_range(batch, rank, world)
        plan = model.plan_for(hi - lo, h, w, 1000, True)
        torch.cuda.synchronize()
        np.savez(os.path.join(debug, 'rank{}.npz'.format(rank)), planes=plan.planes.cpu().numpy(),
                 canon=plan.poll_ws.cpu().numpy().view(np.float32), P_inv=plan.P_inv.cpu().numpy(),
                 boxes=plan.boxes.cpu().numpy(), dims=plan.dimensions.cpu().numpy(), orient=plan.orientations.cpu().numpy(),
                 keypoints=plan.keypoints.cpu().numpy(), best=plan.best_index.cpu().numpy(), residuals=plan.residuals.cpu().numpy())
    dist.barrier()
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
