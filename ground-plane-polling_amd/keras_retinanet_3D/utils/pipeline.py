"""
Host-fed streaming inference: the upload of batch k+1 (raw uint8 frames, an uploader thread and a separate
HIP stream) overlaps the GPU work of batch k.

The reference processes one image at a time and blocks on every predict_on_batch
(/root/reference/keras_retinanet_3D/bin/run_network.py:90-111); this is the production-serving form of
the same loop.  Results are bit-identical to model.predict_on_frames called batch by batch.
"""

import numpy as np

from . import distributed as D


class FramePipeline(object):
    """ pipeline = FramePipeline(model); for outs, scale in pipeline.run(batches): ...

    `batches` yields (frames_u8 (B, H, W, 3), P_inv (B, 4, 3), planes (B, N, 4) or (N, 4)); `run` yields the
    list of 8 NumPy output arrays per batch (+ the resize scale), in order.

    Three things overlap with the kernels of batch k: the upload of batch k+1 (uploader thread + copy stream), the host-side
    enqueue of batch k+1 (the host runs one batch ahead: it only waits for batch k-1's results), and the download of batch
    k-1 (ONE (B, 100, 35) float32 tensor written by gpp_pack_detections at the end of each plan run, 14 KB per image, copied
    by a second stream into pinned memory).  That is what the reference's timer brackets -- feed + run + fetch,
    bin/run_network.py:108-111 -- in its streaming form; the compute stream never waits for the host. """

    def __init__(self, model, depth=4, graph=False, pinned=False, inline=False):
        import torch
        self.model = model
        self.torch = torch
        self.depth = max(3, int(depth))
        # measured on MI355X, 8 frames per batch (tools/pipe_variants.py): resident plan runs 1597 images/s; this pipeline 1525
        # (95.5 %); with graph=True (the plan replayed as one HIP graph launch) 1100; with pinned=True (results copied into a
        # pinned buffer by a third stream) 1097-1479 depending on depth -- re-used pinned host buffers stall this platform's
        # DMA now and then, in either direction -- so both stay off
        self.graph = bool(graph)
        self.pinned = bool(pinned)
        # How the results leave (measured on MI355X with tools/bench_pipeline.py, 8 binary-noise frames per batch, three runs of 60
        # batches each; the plan alone on the same frames resident in HBM: 1621 images/s):
        #   default: a blocking .cpu() of the packed tensor on the compute stream          1594-1606 (98-99 % of resident), steady
        #   inline=True: the compute stream copies into a pinned buffer behind the pack kernel, the host waits for that copy's
        #                event only                                                         1515-1600, occasional 7-8 ms batches
        #   pinned=True: a third stream copies into a pinned buffer                         1471-1560, occasional 20 ms batches
        # (re-used pinned host buffers stall this platform's DMA now and then), so the plain form stays the default
        self.inline = bool(inline)
        self.copy_stream = torch.cuda.Stream()
        self.down_stream = torch.cuda.Stream()
        self.slots = None
        from concurrent.futures import ThreadPoolExecutor
        self.pool = ThreadPoolExecutor(max_workers=1)

    def _make_slots(self, frames, P_inv, planes):
        torch = self.torch
        dev = self.model.device
        self.slots = []
        B = int(frames.shape[0])
        from ..layers.filter_detections import MAX_DETECTIONS as DET
        for _ in range(self.depth):
            self.slots.append({
                'd_frames': torch.empty(tuple(frames.shape), dtype=torch.uint8, device=dev),
                'd_pinv': torch.empty(tuple(P_inv.shape), dtype=torch.float32, device=dev),
                'd_planes': torch.empty(tuple(planes.shape), dtype=torch.float32, device=dev),
                # B x 100 x 35 packed detections + the 8 bytes of the f16x3 range-event counter behind them (model.pack_with_range)
                'd_packed': torch.empty((B * DET * D.PACK_WIDTH + 2,), dtype=torch.float32, device=dev),
                'h_packed': torch.empty((B * DET * D.PACK_WIDTH + 2,), dtype=torch.float32).pin_memory() if (self.pinned or self.inline) else None,
                'B': B,
                'uploaded': torch.cuda.Event(), 'consumed': torch.cuda.Event(), 'done': torch.cuda.Event(),
                'downloaded': torch.cuda.Event(),
            })

    def _upload(self, slot, frames, P_inv, planes):
        """ Runs on the uploader thread.  The copies come straight from the caller's (pageable) arrays through the
        HIP runtime's own staging buffers: they block this thread only (the GIL is released) while the main thread
        enqueues kernels.  (Staging the FRAMES through torch pinned tensors was tried first: re-writing a pinned buffer on
        the CPU and DMA-ing it again stalled the stream for ~85 ms every few batches on this platform.) """
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
        cur.wait_event(slot['downloaded'])                         # d_packed of this slot has been fetched
        plan, scale = self.model.stage_frames(slot['d_frames'], slot['d_pinv'], slot['d_planes'])
        slot['consumed'].record(cur)
        if self.graph and getattr(plan, 'graph', None) is None:
            self.model.capture(plan)
        self.model.run_plan(plan)
        self.model.pack_with_range(plan, slot['d_packed'])         # the plan's buffers are overwritten by the next batch: the results
        #                                                            (and the f16x3 range counter as of this batch) leave through the slot
        slot['done'].record(cur)
        if self.inline:
            slot['h_packed'].copy_(slot['d_packed'], non_blocking=True)
            slot['downloaded'].record(cur)
        elif self.pinned:
            with torch.cuda.stream(self.down_stream):
                self.down_stream.wait_event(slot['done'])
                slot['h_packed'].copy_(slot['d_packed'], non_blocking=True)
                slot['downloaded'].record(self.down_stream)
        slot['scale'] = scale
        slot['plan'] = plan

    def run(self, batches):
        """ Software pipeline over the batches: while batch k is being enqueued (and batches k-1, k-2 run on the GPU), the
        uploader thread already copies batch k+1; results are yielded depth - 2 batches behind the enqueue. """
        it = iter(batches)
        pending = []                    # launched, not yet yielded (oldest first); at most depth - 2 of them
        cap = self.depth - 2

        def start_upload(k, item):
            frames, P_inv, planes = item
            if self.slots is None:
                self._make_slots(np.asarray(frames), np.asarray(P_inv), np.asarray(planes))
                for s in self.slots:
                    s['consumed'].record(self.torch.cuda.current_stream())
                    s['downloaded'].record(self.torch.cuda.current_stream())
            slot = self.slots[k % self.depth]
            return slot, self.pool.submit(self._upload, slot, frames, P_inv, planes)

        item = next(it, None)
        k = 0
        nxt = start_upload(k, item) if item is not None else None
        while nxt is not None:
            slot, fut = nxt
            while len(pending) > cap:
                yield self._collect(pending.pop(0))                 # frees the slot the next upload reuses
            item = next(it, None)
            k += 1
            nxt = start_upload(k, item) if item is not None else None   # overlaps the enqueue below and the kernels in flight
            fut.result()
            self._launch(slot)
            pending.append(slot)
        while pending:
            yield self._collect(pending.pop(0))

    def _collect(self, slot):
        if self.inline or self.pinned:
            slot['downloaded'].synchronize()                        # this batch's copy only: later batches keep running
            flat = slot['h_packed'].numpy().copy()
        else:
            slot['done'].synchronize()
            flat = slot['d_packed'].cpu().numpy()
        outs, count = self.model.unpack_with_range(flat, slot['B'])
        m = self.model
        if m.watches_range() and m.note_range(slot['plan'], count):
            # an activation of THIS batch left the half range (the counter is the plan's own and the batches of a pipeline pass through
            # it in order: whatever it gained since the previous batch's snapshot is this batch's): the batch is run again at float32 from the
            # slot's own inputs, which nothing has overwritten yet (on_range_event) -- on a stream of its own, so that the float32 run does
            # not queue up behind the batches already in flight on the compute stream (they keep running beside it)
            torch = self.torch
            if getattr(self, 'fallback_stream', None) is None:
                self.fallback_stream = torch.cuda.Stream(priority=-1)
            self.fallback_stream.wait_event(slot['uploaded'])
            with torch.cuda.stream(self.fallback_stream):
                outs = m._range_event([slot['d_frames'], slot['d_pinv'], slot['d_planes']], 'predict_on_frames')
        return outs, slot['scale']
