"""
Host-fed streaming inference: the upload of batch k+1 (raw uint8 frames, an uploader thread and a separate
HIP stream) overlaps the GPU work of batch k.

The reference processes one image at a time and blocks on every predict_on_batch
(/root/reference/keras_retinanet_3D/bin/run_network.py:90-111); this is the production-serving form of
the same loop.  Results are bit-identical to model.predict_on_frames called batch by batch.
"""

import numpy as np

from ..backend import hip


class FramePipeline(object):
    """ pipeline = FramePipeline(model, batch, frame_shape, n_planes); for out in pipeline.run(batches): ...

    `batches` yields (frames_u8 (B, H, W, 3), P_inv (B, 4, 3), planes (B, N, 4) or (N, 4)); `run` yields the
    list of 8 NumPy output arrays per batch, in order. """

    def __init__(self, model, depth=2):
        import torch
        self.model = model
        self.torch = torch
        self.depth = int(depth)
        self.copy_stream = torch.cuda.Stream()
        self.slots = None
        from concurrent.futures import ThreadPoolExecutor
        self.pool = ThreadPoolExecutor(max_workers=1)

    def _make_slots(self, frames, P_inv, planes):
        torch = self.torch
        dev = self.model.device
        self.slots = []
        for _ in range(self.depth):
            self.slots.append({
                'd_frames': torch.empty(tuple(frames.shape), dtype=torch.uint8, device=dev),
                'd_pinv': torch.empty(tuple(P_inv.shape), dtype=torch.float32, device=dev),
                'd_planes': torch.empty(tuple(planes.shape), dtype=torch.float32, device=dev),
                'uploaded': torch.cuda.Event(), 'consumed': torch.cuda.Event(), 'done': torch.cuda.Event(),
            })

    def _upload(self, slot, frames, P_inv, planes):
        """ Runs on the uploader thread.  The copies come straight from the caller's (pageable) arrays through the
        HIP runtime's own staging buffers: they block this thread only (the GIL is released) while the main thread
        waits for the GPU.  (Staging through torch pinned tensors was tried first: re-writing a pinned buffer on the
        CPU and DMA-ing it again stalled the stream for ~85 ms every few batches on this platform.) """
        torch = self.torch
        torch.cuda.set_device(self.model.device)
        with torch.cuda.stream(self.copy_stream):
            self.copy_stream.wait_event(slot['consumed'])          # the previous user of this slot has read it
            slot['d_frames'].copy_(torch.as_tensor(np.ascontiguousarray(frames)), non_blocking=True)
            slot['d_pinv'].copy_(torch.as_tensor(np.ascontiguousarray(P_inv, dtype=np.float32)), non_blocking=True)
            slot['d_planes'].copy_(torch.as_tensor(np.ascontiguousarray(planes, dtype=np.float32)), non_blocking=True)
            slot['uploaded'].record(self.copy_stream)

    def _launch(self, slot):
        torch = self.torch
        cur = torch.cuda.current_stream()
        cur.wait_event(slot['uploaded'])
        plan, scale = self.model.stage_frames(slot['d_frames'], slot['d_pinv'], slot['d_planes'])
        slot['consumed'].record(cur)
        self.model.run_plan(plan)
        outs = self.model.outputs(plan)
        slot['outs'] = outs
        slot['done'].record(cur)
        slot['scale'] = scale

    def run(self, batches):
        """ Only ONE batch of kernels is in flight at a time (the HIP runtime was seen to block the host for
        ~85 ms once a few hundred launches are queued); what overlaps with the GPU work of batch k is the upload
        of batch k+1 on the copy stream and the host-side preparation of its arguments. """
        prev = None
        k = 0
        for frames, P_inv, planes in batches:
            if self.slots is None:
                self._make_slots(np.asarray(frames), np.asarray(P_inv), np.asarray(planes))
                for s in self.slots:
                    s['consumed'].record(self.torch.cuda.current_stream())
            slot = self.slots[k % self.depth]
            fut = self.pool.submit(self._upload, slot, frames, P_inv, planes)   # overlaps the kernels of the previous batch
            if prev is not None:
                yield self._collect(prev)                           # host waits for the previous batch here
            fut.result()
            self._launch(slot)
            prev = slot
            k += 1
        if prev is not None:
            yield self._collect(prev)

    def _collect(self, slot):
        slot['done'].synchronize()
        return [o.cpu().numpy() for o in slot['outs']], slot['scale']
