"""
CPU tests of the image-sharded data-parallel path with two gloo processes (world_size 2):
shard ranges, pack/unpack round trip, and that the gathered result equals the single-process
result byte for byte (there is no reduction in the path).
"""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from keras_retinanet_3D.utils import distributed as D


def fake_outputs(batch, seed):
    g = np.random.default_rng(seed)
    f = lambda *s: torch.as_tensor(g.normal(size=s).astype(np.float32))  # noqa: E731
    labels = torch.as_tensor(g.integers(-1, 1, size=(batch, 100)).astype(np.int32))
    orient = torch.as_tensor(g.integers(-1, 4, size=(batch, 100)).astype(np.int32))
    return [f(batch, 100, 12), f(batch, 100, 3), f(batch, 100), labels, orient, f(batch, 100, 4, 3), f(batch, 100, 1, 4), f(batch, 100)]


def test_shard_ranges_cover_the_batch_contiguously():
    for batch in (1, 7, 8, 64, 65):
        for world in (1, 2, 3, 8):
            spans = [D.shard_range(batch, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == batch
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
    assert D.shard_range(64, 3, 8) == (24, 32)            # BASELINE config 3: 8 images per GPU


def test_pack_unpack_round_trip_keeps_dtypes_and_bytes():
    outs = fake_outputs(3, 0)
    packed = D.pack_outputs(outs)
    assert tuple(packed.shape) == (3, 100, D.PACK_WIDTH) and packed.dtype == torch.float32
    back = D.unpack_outputs(packed)
    for a, b in zip(outs, back):
        assert tuple(a.shape) == b.shape and np.array_equal(a.numpy(), b)
    assert back[3].dtype == np.int32 and back[4].dtype == np.int32 and back[0].dtype == np.float32


def _worker(rank, world, port, batch, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    full = D.pack_outputs(fake_outputs(batch, 42))                    # every rank knows the full answer
    lo, hi = D.shard_range(batch, rank, world)
    sizes = [D.shard_range(batch, r, world)[1] - D.shard_range(batch, r, world)[0] for r in range(world)]
    gathered = D.gather_detections(full[lo:hi].contiguous(), sizes)
    q.put((rank, bool(torch.equal(gathered, full)), tuple(gathered.shape)))
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _run(batch):
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, batch, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return results


def test_two_rank_gather_equals_single_process_result_even_shards():
    for rank, same, shape in _run(8):
        assert same and shape == (8, 100, D.PACK_WIDTH)


def test_two_rank_gather_with_ragged_shards():
    for rank, same, shape in _run(5):
        assert same and shape == (5, 100, D.PACK_WIDTH)
