"""
Image-sharded data-parallel inference: one process per GPU, one collective per step.

The reference has no inference-time parallelism at all (its only multi-device code is the training
helper keras.utils.multi_gpu_model, bin/train.py:100-104).  The path shards trivially because every
stage keeps the batch dimension (FilterDetections is a per-image map_fn,
layers/filter_detections.py:257-262; polling is per (image, detection)): rank r of W ranks takes
the contiguous images [r*B/W, (r+1)*B/W), weights and the plane database are replicated, and the
only exchange is ONE all-gather of the packed final detections, 100 x 35 float32 = 14 000 bytes per
image (RCCL over xGMI under torch.distributed backend 'nccl'; latency-bound at this size).
No reduction takes place, and every bit of an image's result is a function of (image, weights, dtype) alone -- block tiles
are tuned per process but never change a result, split-K follows a rule of the layer alone (include/gpp.h,
gpp_conv2d_split_rule) -- so the gathered result is bit-identical to a single-GPU run of the whole batch
(tests/test_zz_sharded_gpu.py: two processes with the real model against one process, byte for byte).
"""

import numpy as np

PACK_WIDTH = 35      # 12 box + 3 dim + score + label + orientation + 12 keypoints + 4 plane + residual
_SLICES = [(0, 12), (12, 15), (15, 16), (16, 17), (17, 18), (18, 30), (30, 34), (34, 35)]
_SHAPES = [(12,), (3,), (), (), (), (4, 3), (1, 4), ()]


def shard_range(global_batch, rank, world_size):
    """ contiguous split; the first (global_batch % world_size) ranks take one extra image """
    base, extra = divmod(int(global_batch), int(world_size))
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def pack_outputs(outputs):
    """ the 8 model outputs (torch tensors) -> one (B, 100, 35) float32 tensor (ints are small: exact) """
    import torch
    b, d = outputs[0].shape[:2]
    if outputs[0].is_cuda and all(o.is_contiguous() for o in outputs):        # one launch of the library's pack kernel
        from ..backend import hip
        out = torch.empty((b, d, PACK_WIDTH), dtype=torch.float32, device=outputs[0].device)
        hip.check(hip.lib().gpp_pack_detections(*([hip.ptr(o) for o in outputs] + [int(b), int(d), hip.ptr(out), hip.stream_ptr()])),
                  'gpp_pack_detections')
        return out
    return torch.cat([o.reshape(b, d, -1).to(torch.float32) for o in outputs], dim=2).contiguous()


def unpack_outputs(packed):
    """ inverse of pack_outputs; NumPy arrays with the reference's dtypes (labels / orientations int32) """
    import torch
    arr = packed.cpu().numpy() if isinstance(packed, torch.Tensor) else np.asarray(packed)
    out = []
    for k, ((lo, hi), shp) in enumerate(zip(_SLICES, _SHAPES)):
        a = np.ascontiguousarray(arr[:, :, lo:hi]).reshape(arr.shape[:2] + shp)
        out.append(a.astype(np.int32) if k in (3, 4) else a.astype(np.float32))
    return out


def gather_detections(packed_local, shard_sizes=None, group=None, async_op=False):
    """ all-gather the packed detections of every rank -> (B_global, 100, 35) on every rank.
    async_op (equal shards on RCCL only): returns (tensor, work) without making the caller's stream wait for the
    collective, so that the next step's kernels run while the gather is on the wire; call work.wait() before reading. """
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return (packed_local, None) if async_op else packed_local
    world = dist.get_world_size(group)
    if shard_sizes is None:
        shard_sizes = [packed_local.shape[0]] * world
    equal = len(set(shard_sizes)) == 1
    if equal and dist.get_backend(group) == 'nccl':
        out = torch.empty((sum(shard_sizes),) + tuple(packed_local.shape[1:]), dtype=packed_local.dtype, device=packed_local.device)
        work = dist.all_gather_into_tensor(out, packed_local, group=group, async_op=async_op)
        return (out, work) if async_op else out
    if async_op:
        raise ValueError('async gather needs equal shards on the nccl (RCCL) backend')
    pad = max(shard_sizes)
    device = packed_local.device
    local = packed_local.cpu() if dist.get_backend(group) == 'gloo' else packed_local      # gloo exchanges host tensors
    if local.shape[0] < pad:
        local = torch.cat([local, local.new_zeros((pad - local.shape[0],) + tuple(local.shape[1:]))], dim=0)
    chunks = [torch.empty_like(local) for _ in range(world)]
    dist.all_gather(chunks, local.contiguous(), group=group)
    return torch.cat([c[:n] for c, n in zip(chunks, shard_sizes)], dim=0).to(device)


class ShardedModel(object):
    """ Wraps a model: predict_on_batch on a GLOBAL batch, each rank computing its contiguous shard. """

    def __init__(self, model, group=None):
        self.model = model
        self.group = group

    def predict_on_batch(self, inputs):
        import torch.distributed as dist
        images, P_inv, planes = inputs
        world = dist.get_world_size(self.group) if dist.is_initialized() else 1
        rank = dist.get_rank(self.group) if dist.is_initialized() else 0
        B = images.shape[0]
        lo, hi = shard_range(B, rank, world)
        local_planes = planes[lo:hi] if len(planes.shape) == 3 else planes
        plan = self.model.stage_inputs([images[lo:hi], P_inv[lo:hi], local_planes])
        self.model.run_plan(plan)
        packed = self._packed_shard(plan, hi - lo)
        sizes = [shard_range(B, r, world)[1] - shard_range(B, r, world)[0] for r in range(world)]
        return unpack_outputs(gather_detections(packed, sizes, self.group))

    def _packed_shard(self, plan, n_local):
        """ this rank's (n_local, 100, 35) packed detections.  A dtype='f16x3' model watches the half range here as every synchronous call does
        (models/retinanet.py `on_range_event`): the counter rides behind the packed shard, is read BEFORE the shard goes on the wire, and a
        shard whose activations left the range is recomputed on the float32 twin first -- the gathered batch never holds a clamped result. """
        model = self.model
        if not (hasattr(model, 'watches_range') and model.watches_range()):
            return pack_outputs(model.outputs(plan))
        import torch
        flat = model.pack_with_range(plan)
        det = int(model.outputs(plan)[0].shape[1])                # detections per image (layers.filter_detections.MAX_DETECTIONS)
        n = n_local * det * PACK_WIDTH
        count = int(flat[n:].cpu().numpy().view(np.uint64)[0])                       # 8 bytes; the stream has reached the end of this rank's plan
        if not model.note_range(plan, count):
            return flat[:n].view(n_local, det, PACK_WIDTH)
        outs = model._range_event([plan.images, plan.P_inv, plan.planes], 'predict_on_batch')
        host = np.concatenate([np.asarray(o, np.float32).reshape(n_local, det, -1) for o in outs], axis=2)
        return torch.as_tensor(np.ascontiguousarray(host)).to(flat.device)
