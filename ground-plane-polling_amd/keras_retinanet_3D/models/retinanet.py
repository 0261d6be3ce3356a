"""
RetinaNet-3D inference model on MI355X: the object `models.load_model` returns.

It plays the role of the keras.models.Model that the reference builds in
/root/reference/keras_retinanet_3D/models/retinanet.py:359-422 (`retinanet_bbox`):
    backbone (keras_resnet, models/resnet.py:88-102) -> C3, C4, C5
    __create_pyramid_features :170-205                -> P3..P7 (512 channels)
    regression / regression_dim / classification heads :24-167, applied per level, concatenated :257-281
    Anchors + RegressBoxes + RegressDims :284-311, :411-412 ; FilterDetections :415 ; FitRoadPlanes :416
with the same `predict_on_batch([images, P_inv, planes])` -> 8 arrays contract
(bin/run_network.py:105-110; output order retinanet.py:418-419).

Nothing is traced or compiled at run time: for a given (batch, H, W) the model lays out its
activations in HBM once and records a *plan* -- a flat array of C-ABI descriptors (include/gpp.h) --
that one `gpp_plan_run` call enqueues on the current HIP stream.  All arithmetic happens in the
hand-written kernels of ../csrc; PyTorch only owns the device buffers.

HBM layout
  * activations NHWC, 16-bit (bf16 default, f16) or float32 (dtype='f32': the reference's own arithmetic type --
    float32 operands on v_mfma_f32_16x16x4_f32, float32 stem, no fused bottleneck tails; dtype='bf16x3': the same float32
    storage with every float32 product computed as three bf16 matrix products, ~2^-16 relative), one dense buffer per live tensor
  * the five pyramid levels of every FPN / head tensor are stored back to back per image,
    (B, 11438, C) for a 402x1333 input, so that one grouped launch covers all levels and the head
    outputs come out directly in the reference's concatenated (B, A, k) order
  * head outputs (classification logits, fused 144-channel regression, dimensions) float32
  * weights [C_out][KH*KW*C_in] in the storage type with the frozen BatchNormalization folded in, biases float32
Every bit of a result is a function of (image, weights, dtype) alone: block tiles are tuned by timing but never change
a result, and split-K follows a rule of the layer alone (gpp_conv2d_split_rule) -- not of the batch size or the rank.
"""

import ctypes
import json
import os

import numpy as np

from ..backend import hip
from ..layers import conv as C
from ..layers.filter_detections import MAX_DETECTIONS, NMS_THRESHOLD, SCORE_THRESHOLD
from ..utils import anchors as anchor_utils
from ..utils.gpp_utils import POLL_THRESHOLD
from . import weights as W


class StemDesc(ctypes.Structure):
    _fields_ = [('inp', ctypes.c_void_p), ('weight', ctypes.c_void_p), ('bias', ctypes.c_void_p), ('out', ctypes.c_void_p),
                ('dtype', ctypes.c_int32), ('B', ctypes.c_int32), ('H', ctypes.c_int32), ('W', ctypes.c_int32), ('range_counter', ctypes.c_void_p)]


class PoolDesc(ctypes.Structure):
    _fields_ = [('inp', ctypes.c_void_p), ('out', ctypes.c_void_p), ('dtype', ctypes.c_int32), ('B', ctypes.c_int32),
                ('H', ctypes.c_int32), ('W', ctypes.c_int32), ('C', ctypes.c_int32), ('reserved', ctypes.c_int32)]


class ReluDesc(ctypes.Structure):
    _fields_ = [('inp', ctypes.c_void_p), ('out', ctypes.c_void_p), ('in_bstride', ctypes.c_int64),
                ('out_bstride', ctypes.c_int64), ('count', ctypes.c_int64), ('dtype', ctypes.c_int32), ('B', ctypes.c_int32)]


class DetectDesc(ctypes.Structure):
    _fields_ = [(n, ctypes.c_void_p) for n in ('cls_logits', 'regression', 'regression_dim', 'anchors', 'boxes', 'dims',
                                               'scores', 'labels', 'orientations', 'anchor_index', 'counts', 'workspace')] + \
               [('workspace_bytes', ctypes.c_size_t), ('n_anchors', ctypes.c_int64),
                ('B', ctypes.c_int32), ('num_base_anchors', ctypes.c_int32), ('fused_layout', ctypes.c_int32),
                ('max_det', ctypes.c_int32), ('score_thr', ctypes.c_float), ('iou_thr', ctypes.c_float)]


class PollDesc(ctypes.Structure):
    _fields_ = [(n, ctypes.c_void_p) for n in ('boxes', 'dims', 'orient', 'P_inv', 'planes', 'keypoints', 'keyplanes',
                                               'residuals', 'best_idx', 'workspace')] + \
               [('workspace_bytes', ctypes.c_size_t), ('B', ctypes.c_int32), ('D', ctypes.c_int32), ('N', ctypes.c_int32),
                ('planes_batched', ctypes.c_int32), ('thr', ctypes.c_float), ('reserved', ctypes.c_int32)]


class PlanOp(ctypes.Structure):
    _fields_ = [('kind', ctypes.c_int32), ('tag', ctypes.c_int32), ('desc', ctypes.c_void_p)]


class TailDesc(ctypes.Structure):
    _fields_ = [('conv3x3', ctypes.c_void_p), ('conv1x1', ctypes.c_void_p), ('tile_rows', ctypes.c_int32), ('reserved', ctypes.c_int32)]


class BlockDesc(ctypes.Structure):
    _fields_ = [('conv1x1_a', ctypes.c_void_p), ('conv3x3_b', ctypes.c_void_p), ('conv1x1_c', ctypes.c_void_p), ('tile', ctypes.c_int32), ('reserved', ctypes.c_int32)]


OP_STEM, OP_MAXPOOL, OP_CONV, OP_RELU, OP_DETECT, OP_POLL, OP_TAIL = 1, 2, 3, 4, 5, 6, 7
OP_BLOCK = 16
OP_DETECT_CANDIDATES, OP_DETECT_SELECT, OP_DETECT_EMIT = 8, 9, 10
OP_DETECT_OSF = 12
OP_STEM_POOL = 13
DETECT_OPS = (OP_DETECT, OP_DETECT_CANDIDATES, OP_DETECT_SELECT, OP_DETECT_EMIT, 12)
OP_JOIN, OP_SYNC = 0x10000, 0x20000


class Plan(object):
    """ Everything one (batch, H, W, N planes) configuration needs: buffers, descriptors, op array. """

    def __init__(self):
        self.keep = []          # ctypes descriptors and torch buffers kept alive
        self.io = {}            # conv op name -> (input FMaps, output FMaps, residual FMaps or None); half-batch plans: the LAST part
        self.io_parts = {}      # conv op name -> [(inputs, outputs, residuals) of every launch under that name] (half-batch plans: two)
        self.tuning_parts = {}  # conv op name -> [(tile, us) of every launch under that name]
        self.ops = []           # (kind, tag, desc, name, flops)
        self.oracle_names = {}  # fused ops: reference layer name of each output map (per-layer parity tests)
        self.lanes = []         # per op: side-stream lane << 8 | join flag (include/gpp.h GPP_OP_LANE / GPP_OP_JOIN)
        self.op_batch = {}      # id(descriptor) -> images the launch covers where that is not the plan's batch (half-batch launches)
        self.access = {}        # id(descriptor) -> (byte intervals read, byte intervals written): check_stream_ordering
        self.array = None
        self.flops = 0.0

    def add(self, kind, desc, name, tag=0, flops=0.0, lane=0, join=False, sync=False):
        self.keep.append(desc)
        self.ops.append((kind, tag, desc, name, flops))
        self.lanes.append((int(lane) << 8) | (OP_JOIN if join else 0) | (OP_SYNC if sync else 0))
        self.flops += flops

    # ---- who reads and writes what: the byte intervals (one per image) every launch touches, by descriptor.  check_stream_ordering()
    # replays gpp_plan_run's fork / join rules over them: the plan builder places launches on side streams by hand, and a missing
    # join is a race that shows up once in a while, at full size only (round 4 found one between a split and an unsplit stage)
    @staticmethod
    def span(fm, esz=None):
        """ byte intervals of an FMap, one per image """
        e = fm.buf.element_size() if esz is None else esz
        base = fm.buf.data_ptr() + fm.off * e
        size = ((fm.H * fm.W - 1) * fm.pitch + fm.C) * e
        return [(base + b * fm.bstride * e, base + b * fm.bstride * e + size) for b in range(fm.B)]

    @staticmethod
    def span_of(tensor):
        return [(tensor.data_ptr(), tensor.data_ptr() + tensor.numel() * tensor.element_size())]

    def touch(self, desc, reads=(), writes=(), kind=None):
        """ kind: for descriptors shared by several ops (the three decode stages), the op kind the accesses belong to """
        self.access[id(desc) if kind is None else (id(desc), kind)] = ([iv for r in reads for iv in r], [iv for w in writes for iv in w])

    def check_stream_ordering(self):
        """ every pair of launches on DIFFERENT streams that touch overlapping bytes (at least one of them writing) must be ordered by a
        fork or a join, as gpp_plan_run (csrc/plan.cpp) places them: a side-lane launch forks from the caller's stream when its lane is
        not open (or carries SYNC); a JOIN launch on the caller's stream (and the end of the plan) closes every open lane.
        Returns the list of violations [(earlier op, later op)], empty when the plan is race-free by construction. """
        def overlap(a, b):
            return any(x0 < y1 and y0 < x1 for x0, x1 in a for y0, y1 in b)
        bad, seen = [], []                    # seen: (position, lane, name, reads, writes)
        active = {}
        forks, joins = {}, []                 # lane -> positions of its forks; positions of joins
        for pos, (kind, _, desc, name, _) in enumerate(self.ops):
            flags = self.lanes[pos]
            lane, join, sync = (flags >> 8) & 0xff, bool(flags & OP_JOIN), bool(flags & OP_SYNC)
            if lane > 0:
                if not active.get(lane) or sync:
                    forks.setdefault(lane, []).append(pos)
                    active[lane] = True
            elif join:
                joins.append(pos)
                active = {}
            reads, writes = self.access.get((id(desc), kind)) or self.access.get(id(desc), ((), ()))
            for p0, l0, n0, r0, w0 in seen:
                if l0 == lane or not (overlap(w0, reads) or overlap(w0, writes) or overlap(r0, writes)):
                    continue
                j = [q for q in joins if p0 < q <= pos]                      # a join after the earlier launch, not after this one
                if l0 == 0:
                    ok = any(p0 < f <= pos for f in forks.get(lane, []))     # this lane forked after the main-stream launch
                elif lane == 0:
                    ok = bool(j)
                else:
                    ok = bool(j) and any(min(j) <= f <= pos for f in forks.get(lane, []))
                if not ok:
                    bad.append((n0, name))
            seen.append((pos, lane, name, reads, writes))
        return bad

    @staticmethod
    def stage_of(kind, name):
        """ include/gpp.h GPP_OP_STAGE: 1 stem, 2 backbone, 3 FPN, 4 heads, 5 decode, 6 polling (roctx ranges under GPP_ROCTX=1) """
        if kind in (OP_STEM, OP_MAXPOOL, OP_STEM_POOL):
            return 1
        if kind in DETECT_OPS:
            return 5
        if kind == OP_POLL:
            return 6
        if name.startswith('res'):
            return 2
        if name.startswith('pyramid_'):
            return 4
        return 3                                  # C5_reduced ... P3, P6, C6_relu, P7

    def finalize(self):
        arr = (PlanOp * len(self.ops))()
        for i, (kind, tag, desc, name, _) in enumerate(self.ops):
            arr[i].kind, arr[i].tag, arr[i].desc = kind | self.lanes[i] | (self.stage_of(kind, name) << 20), tag, ctypes.addressof(desc)
        self.array = arr


class RetinaNet3D(object):
    """ Inference model: ResNet-50/101/152 + FPN + heads + decode + ground-plane polling. """

    def __init__(self, weights, backbone_name='resnet50', dtype='f16x3', nms=True, class_specific_filter=True,
                 orientation_specific_filter=False, name='retinanet-bbox', on_range_event=None, plan=None):
        import torch
        # 'throughput' | 'latency' (models.load_model): which layers split their K loop (layers/conv.latency_split)
        self.plan_mode = plan or os.environ.get('GPP_PLAN', 'throughput')
        if self.plan_mode not in ('throughput', 'latency'):
            raise ValueError("plan must be 'throughput' or 'latency', got {!r}".format(self.plan_mode))
        # dtype='f16x3' only -- what happens when an activation of a call left the IEEE-half range (a finite value beyond +-65504 is
        # clamped when it is split: a plausible wrong answer; the epilogues count such stores, gpp_x3_range_events):
        #   'f32' (default)  the call is run again at dtype='f32' (a float32 twin of the model, built at the first event) and THAT result
        #                    is returned: the drop-in returns what the reference's floatx graph returns, slower for that call
        #   'raise'          GppError
        #   'ignore'         rounds 3-4: the counter is only there to be read (model.x3_range_events())
        # The counter is read with the results a synchronous call fetches anyway (fetch(), FramePipeline): no extra synchronisation.
        self.on_range_event = on_range_event or os.environ.get('GPP_ON_RANGE_EVENT', 'f32')
        if self.on_range_event not in ('f32', 'raise', 'ignore'):
            raise ValueError("on_range_event must be 'f32', 'raise' or 'ignore', got {!r}".format(self.on_range_event))
        self.range_fallbacks = 0         # calls whose result was replaced (or refused) because of a range event
        self._twin = None
        self._weights = weights if (dtype == 'f16x3' and self.on_range_event == 'f32') else None
        self.class_specific_filter = class_specific_filter
        self.osf = bool(orientation_specific_filter)     # per-orientation NMS (filter_detections.py:84-98), gpp_detect_osf_f32
        self.nms = bool(nms)
        self.name = name
        self.backbone_name = backbone_name.split('_')[0]
        if self.backbone_name not in W.BLOCKS:
            raise ValueError('Backbone (\'{}\') not in allowed backbones ({}).'.format(backbone_name, sorted(W.BLOCKS)))
        if dtype not in ('bf16', 'f16', 'f32', 'bf16x3', 'f16x3'):
            raise ValueError("dtype must be 'bf16', 'f16', 'f32', 'bf16x3' or 'f16x3', got {!r}".format(dtype))
        self.dtype = dtype
        self.esz = C.elem_size(dtype)
        self.tdtype = C.torch_dtype(dtype)
        self.device = hip.require_device()
        hip.lib()
        self.torch = torch
        self._plans = {}
        self._anchors = {}
        self._tuned = {}             # (layer, B, H, W) -> (tile_hint, split_k, us): see _autotune
        self._load_tune_cache()
        self._upload(weights)
        self.tag_names = []          # filled by the plan builder: names of event-tagged ops

    # ------------------------------------------------------------------ weights
    def _upload(self, weights):
        torch, dev = self.torch, self.device
        W.validate_weights(weights, self.backbone_name)
        self.conv_w = {}
        self.conv_scale = {}

        def put(name, kernel, bias):
            self.conv_w[name] = (C.pack_weight(kernel, self.dtype, dev), torch.as_tensor(bias).to(dev).contiguous(),
                                 kernel.shape)
            if self.dtype == 'f16x3':        # the inverse of the per-channel power of two the packed weights carry (layers/conv.py)
                self.conv_scale[name] = C.out_scale_of(kernel, dev)

        for conv, bn, kh, kw, cin, cout, _ in W.backbone_layers(self.backbone_name):
            k, b = W.folded_conv(weights, conv, bn)
            if k.shape != (kh, kw, cin, cout):
                raise ValueError('weight {} has shape {}, expected {}'.format(conv, k.shape, (kh, kw, cin, cout)))
            if conv == 'conv1':
                if self.dtype in C.X3_TYPES and os.environ.get('GPP_X3_STEM', 'mfma') != 'valu':
                    # the x3 types: conv1 on the matrix pipe, input and weights split into two IEEE halves (csrc/stem.hip)
                    self.stem_w = hip.pack_stem_weights_x3(k.reshape(147, 64), dev)
                    self.stem_x3 = True
                elif self.esz == 4:          # float32 stem on the vector ALUs: the folded kernel as it is, [147][64]
                    self.stem_w = torch.as_tensor(np.ascontiguousarray(k.reshape(147, 64), dtype=np.float32)).to(dev).contiguous()
                else:
                    self.stem_w = hip.pack_stem_weights(k.reshape(147, 64), dev)
                self.stem_b = torch.as_tensor(b).to(dev).contiguous()
            else:
                put(conv, k, b)
        for name, k_, cin, cout, _ in W.fpn_layers():
            k, b = W.folded_conv(weights, name)
            put(name, k, b)
        for name, cin, cout, kind in W.head_layers():
            if name.startswith('pyramid_regression_op'):
                continue
            k, b = W.folded_conv(weights, name)
            put(name, k, b)
        k, b = W.fused_regression_outputs(weights)
        put('pyramid_regression_ops', k, b)
        k, b = W.fused_tower_inputs(weights)
        put('pyramid_towers_0', k, b)

    # ------------------------------------------------------------------ plan
    def _anchor_table(self, hw):
        if hw not in self._anchors:
            self._anchors[hw] = self.torch.as_tensor(anchor_utils.anchors_for_image(hw)).to(self.device).contiguous()
        return self._anchors[hw]

    def _desc(self, plan, name, inputs, outputs, K, stride=1, pad=None, relu=False, residuals=None, out_f32=False, lane=0):
        wt, bias, shape = self.conv_w[name]
        kh, kw, cin, cout = shape
        if pad is None:
            pad = (0, 0)
        # plan='latency': an explicit split-K factor per layer (a function of the layer alone); 0 = the library's own rule
        split = C.latency_split(kh, kw, cin, cout, sum(f.H * f.W for f in outputs)) if self.plan_mode == 'latency' else 0
        d = C.conv_desc(inputs, outputs, wt, bias, kh, kw, cin, cout, stride=stride, pad=pad, relu=relu,
                        residuals=residuals, dtype=self.dtype, out_f32=out_f32, out_scale=self.conv_scale.get(name), split_k=split)
        if self.dtype == 'f16x3':         # this plan's own range-event slot (Plan.range_slot): what its launches count no other plan sees
            d.range_counter = plan.range_slot.data_ptr()
        # split-K partial tiles: the workspace of this op's stream lane is allocated once every op is known (_build)
        plan.conv_descs.append((d, lane))
        plan.ws_need[lane] = max(plan.ws_need.get(lane, 0), C.workspace_bytes(d))
        return d

    def _conv(self, plan, name, inputs, outputs, K, stride=1, pad=None, relu=False, residuals=None, out_f32=False, tag=0, lane=0,
              join=False, sync=False):
        d = self._desc(plan, name, inputs, outputs, K, stride, pad, relu, residuals, out_f32, lane)
        plan.op_batch[id(d)] = inputs[0].B
        plan.add(OP_CONV, d, name, tag=tag, flops=C.conv_flops(d), lane=lane, join=join, sync=sync)
        plan.io[name] = (inputs, outputs, residuals)          # FMaps per op (introspection: per-layer parity tests)
        plan.io_parts.setdefault(name, []).append((inputs, outputs, residuals))
        plan.touch(d, [Plan.span(f) for f in list(inputs) + list(residuals or [])], [Plan.span(f, 4 if out_f32 else None) for f in outputs])

    def _tail(self, plan, nm, a, y, shortcut, join=False, lane=0):
        """ branch2b (3x3) + branch2c (1x1, + shortcut, ReLU) of one bottleneck as ONE launch
        (gpp_bottleneck_tail): the intermediate map never reaches HBM.  Bit-identical to the two layers. """
        d1 = self._desc(plan, 'res{}_branch2b'.format(nm), [a], [a], 3, pad=(1, 1), relu=True)       # its `out` is never written
        d2 = self._desc(plan, 'res{}_branch2c'.format(nm), [a], [y], 1, relu=True, residuals=[shortcut])
        plan.keep += [d1, d2]
        t = TailDesc(ctypes.addressof(d1), ctypes.addressof(d2), 0, 0)
        plan.op_batch[id(t)] = a.B
        name = 'res{}_branch2b+2c'.format(nm)
        plan.add(OP_TAIL, t, name, flops=C.conv_flops(d1) + C.conv_flops(d2), join=join, lane=lane)
        plan.io[name] = ([a], [y], [shortcut])
        plan.io_parts.setdefault(name, []).append(([a], [y], [shortcut]))
        plan.touch(t, [Plan.span(a), Plan.span(shortcut)], [Plan.span(y)])

    def _block(self, plan, nm, x, a, bmap, y, shortcut, stride=1, join=False, lane=0):
        """ a whole bottleneck -- branch2a, branch2b, branch2c (+ shortcut, ReLU) -- as ONE launch (gpp_bottleneck_block): neither intermediate map
        reaches HBM (`a` / `bmap` only lend the descriptors their shapes).  Bit-identical to the three layers. """
        d1 = self._desc(plan, 'res{}_branch2a'.format(nm), [x], [a], 1, stride=stride, relu=True)
        d2 = self._desc(plan, 'res{}_branch2b'.format(nm), [a], [bmap], 3, pad=(1, 1), relu=True)
        d3 = self._desc(plan, 'res{}_branch2c'.format(nm), [bmap], [y], 1, relu=True, residuals=[shortcut])
        plan.keep += [d1, d2, d3]
        t = BlockDesc(ctypes.addressof(d1), ctypes.addressof(d2), ctypes.addressof(d3), 0, 0)
        plan.op_batch[id(t)] = x.B
        name = 'res{}_branch2a+2b+2c'.format(nm)
        plan.add(OP_BLOCK, t, name, flops=C.conv_flops(d1) + C.conv_flops(d2) + C.conv_flops(d3), join=join, lane=lane)
        plan.io[name] = ([x], [y], [shortcut])
        plan.io_parts.setdefault(name, []).append(([x], [y], [shortcut]))
        plan.touch(t, [Plan.span(x), Plan.span(shortcut)], [Plan.span(y)])

    def _build(self, B, H, Wd, n_planes, planes_batched):
        torch, dev, dt = self.torch, self.device, self.tdtype
        plan = Plan()
        plan.shape = (B, H, Wd, n_planes, planes_batched)
        # split-K partial tiles of the deep-K layers with a tiny per-image grid (res5 branch2b, P5..P7); one workspace per
        # stream lane (concurrent launches must not share partial tiles), sized from the descriptors at the end of _build
        head_lanes = os.environ.get('GPP_HEAD_LANES', '0') != '0'
        # res2 .. res5 run as two half batches, the second half on a side stream beside the first (GPP_HALF_LANES="" for whole batches).  res2 joined the
        # list in round 6: with its identity blocks as one two-per-CU launch each (gpp_bottleneck_block) two half-batch chains overlap the HBM-bound phases
        # of one with the matrix phase of the other -- same box, alternating: B = 8 812.8 -> 818.7 images/s (+0.7 %), B = 4 +1.1 %, B = 2 +1.1 %
        # (profiles/r6/ab_half_lanes_with_blocks.txt); before (rounds 4 - 5, separate launches) it lost
        half_stages = set(int(v) for v in os.environ.get('GPP_HALF_LANES', '0,1,2,3').split(',') if v.strip()) if B >= 2 else set()
        br1_lane = os.environ.get('GPP_BR1_LANE', '1') != '0'         # measured +0.4 % on the f16x3 step (same box, alternating)
        plan.conv_descs, plan.ws_need = [], {}
        # dtype='f16x3': the 8-byte counter every launch of THIS plan adds its range events to (gpp_conv_desc.range_counter, gpp_stem_desc.range_counter),
        # never reset by anyone but x3_range_events(reset=True) of this model; range_seen = its value when a result of the plan was last fetched
        plan.range_slot = torch.zeros((1,), dtype=torch.int64, device=dev)
        plan.range_seen = 0

        def fmap(h, w, c, dtype=None):
            f = C.FMap.empty(B, h, w, c, dtype or dt, dev, half=self.dtype if self.dtype in C.X3_TYPES else 'bf16x3')
            plan.keep.append(f.buf)
            return f

        # dtype='bf16x3': maps written and read by convolutions only are stored PRE-SPLIT ([32 bf16 hi | 32 bf16 lo] per 32 channels,
        # gpp_conv_desc.x3_split): GPP_X3_SPLIT=2 (default) every such map, 1 = only the maps between the FPN / head layers, 0 = none
        x3_level = int(os.environ.get('GPP_X3_SPLIT', '2')) if self.dtype in C.X3_TYPES else 0

        def bmap(h, w, c):
            f = fmap(h, w, c)
            return f.mark_split() if x3_level >= 2 else f

        # ---- inputs
        plan.images = torch.empty((B, H, Wd, 3), dtype=torch.float32, device=dev)
        plan.P_inv = torch.empty((B, 4, 3), dtype=torch.float32, device=dev)
        plan.planes = torch.empty((B, n_planes, 4) if planes_batched else (n_planes, 4), dtype=torch.float32, device=dev)

        # ---- stem: conv1 + bn_conv1 + relu, pool1
        H1, W1 = (H + 6 - 7) // 2 + 1, (Wd + 6 - 7) // 2 + 1
        H2, W2 = (H1 + 1) // 2, (W1 + 1) // 2
        x = fmap(H2, W2, 64)
        stem_x3 = getattr(self, 'stem_x3', False)
        if (self.esz == 2 or stem_x3) and os.environ.get('GPP_FUSE_STEM_POOL', '1') != '0':
            # conv1 + bn_conv1 + relu + pool1 in one launch, the (B, H1, W1, 64) conv map is never stored (bit-identical to the two launches:
            # tests/test_stem_gpu.py).  16-bit types since round 2; the x3 types since round 6 (gpp_stem_pool_fused_x3: the float32 conv map was
            # 274 MB written + read back at B = 8)
            d = StemDesc(plan.images.data_ptr(), self.stem_w.data_ptr(), self.stem_b.data_ptr(), x.buf.data_ptr(),
                         hip.GPP_F16X3 if stem_x3 else C.gpp_storage_dtype(self.dtype), B, H, Wd,
                         plan.range_slot.data_ptr() if self.dtype == 'f16x3' else None)
            plan.add(OP_STEM_POOL, d, 'conv1+pool1', flops=2.0 * B * H1 * W1 * 147 * 64)
            plan.touch(d, [Plan.span_of(plan.images)], [Plan.span(x)])
            plan.stem_out = None
        else:
            stem = fmap(H1, W1, 64)
            d = StemDesc(plan.images.data_ptr(), self.stem_w.data_ptr(), self.stem_b.data_ptr(), stem.buf.data_ptr(),
                         hip.GPP_F16X3 if stem_x3 else C.gpp_storage_dtype(self.dtype), B, H, Wd,
                         plan.range_slot.data_ptr() if self.dtype == 'f16x3' else None)
            plan.add(OP_STEM, d, 'conv1', flops=2.0 * B * H1 * W1 * 147 * 64)
            plan.touch(d, [Plan.span_of(plan.images)], [Plan.span(stem)])
            plan.stem_out = stem
            pool_d = PoolDesc(stem.buf.data_ptr(), x.buf.data_ptr(), C.gpp_storage_dtype(self.dtype), B, H1, W1, 64, 0)
            plan.add(OP_MAXPOOL, pool_d, 'pool1')
            plan.touch(pool_d, [Plan.span(stem)], [Plan.span(x)])
        plan.pool_out = x

        # ---- bottleneck stages (keras_resnet bottleneck_2d: stride on the first 1x1).
        # A stage can be run chunk of images by chunk of images (GPP_STAGE_CHUNKS="2,4,8,8") to keep a chunk's
        # working set inside the 256 MiB Infinity Cache; measured on MI355X at B = 8 this LOSES 1-6 % (the smaller
        # launches cost more than the on-die re-reads save), so the default is the whole batch per launch.
        env_chunks = os.environ.get('GPP_STAGE_CHUNKS')
        # widths whose branch2b + branch2c run as one launch (GPP_FUSE_TAIL=0 for none).  Measured at B = 8, same box,
        # whole step: none 1553, res3 only 1562, res2 + res3 1571 images/s (in isolation the fused res2 launch is no
        # faster than its two layers -- 122 us vs 36 + 80 -- but the step is: 69 MB less through HBM per block)
        fuse_tail = [int(v) for v in os.environ.get('GPP_FUSE_TAIL', '64,128').split(',') if v.strip() and int(v) > 0]
        if self.dtype in C.X3_TYPES:
            # the x3 form of the fused tail (bottleneck_tail_x3_kernel): pre-split maps, C = 64 (res2) only -- at C = 128 its LDS
            # footprint leaves one workgroup per CU
            fuse_tail = [v for v in fuse_tail if v == 64] if x3_level >= 2 else []
        elif self.esz == 4:
            fuse_tail = []              # float32 operands: no fused tail
        # widths whose IDENTITY blocks (branch2a + 2b + 2c + shortcut) run as one launch (gpp_bottleneck_block; x3 types on pre-split maps): res2 (C = 64:
        # 4-wavefront workgroups, two per CU; x in once, y out once: 245 -> 212 us per block at B = 8) and res3 (C = 128: 8 wavefronts, one per CU;
        # at parity with its three launches in isolation, half their fabric bytes).  Same-box A/B of the step (profiles/r6/ab_fuse_block_*.txt):
        # B = 8: 780 -> 793 images/s (+1.7 %) with both, +0.9 % with res2 alone; B = 4 +1.2 %, B = 2 +0.7 %, batch-1 plan 2.67 -> 2.65 ms.
        # GPP_FUSE_BLOCK="" for the separate launches (bit-identical either way).  Projection blocks (GPP_FUSE_BLOCK_PROJ=1) measured -0.6 %: off.
        fuse_block = [int(v) for v in os.environ.get('GPP_FUSE_BLOCK', '64,128').split(',') if v.strip()] if (self.dtype in C.X3_TYPES and x3_level >= 2) else []
        fuse_block_proj = os.environ.get('GPP_FUSE_BLOCK_PROJ', '0') != '0'      # block 0 of a stage too (its shortcut is the projection launch's map)

        def sub(fm, c0, nb):
            return C.FMap(fm.buf, nb, fm.H, fm.W, fm.C, off=fm.off + c0 * fm.bstride, bstride=fm.bstride, pitch=fm.pitch, split=fm.split, half=fm.half)

        feats = []
        lane_open = False            # a stage that ran as two half batches leaves its second half on side lane 1: whatever reads the whole
        #                              batch next (a stage that is NOT split, or the FPN) has to join it first
        for stage, n_blocks in enumerate(W.BLOCKS[self.backbone_name]):
            f = 64 * 2 ** stage
            blocks = []
            xin = x
            for block in range(n_blocks):
                nm = W.block_name(self.backbone_name, stage, block)
                stride = 2 if (block == 0 and stage > 0) else 1
                ho, wo = (x.H - 1) // stride + 1, (x.W - 1) // stride + 1
                fused = f in fuse_tail and f in (64, 128)
                rec = {'nm': nm, 'stride': stride, 'a': bmap(ho, wo, f), 'b': None if fused else bmap(ho, wo, f),
                       'sc': bmap(ho, wo, 4 * f) if block == 0 else None, 'y': bmap(ho, wo, 4 * f)}
                blocks.append(rec)
                x = rec['y']
            chunk = max(1, min(B, int(env_chunks.split(',')[stage]))) if env_chunks else B
            # GPP_HALF_LANES (default "0,1,2,3" = res2 .. res5): the stage as two half batches, the second half on a side stream beside
            # the first.  A launch of these stages fills the 256 CUs 0.7 - 1.4 times and is bound by tile fills and first-touch latency;
            # two independent chains in flight overlap one's prologue / epilogue / barrier waits with the other's main loop (f16x3 step,
            # same box, alternating: 775 -> 793 images/s).  An image's result does not depend on its batch (section 4.4): same bytes.
            # The halves stay apart until the FPN's first layer joins them; the per-block shortcut stream is not used inside them.
            halves = stage in half_stages and B >= 2 and chunk >= B       # (a stage that is chunked runs its chunks one after the other)
            # (three parts on three streams, measured: 801 against 811 images/s for two)
            parts = [(B // 2, B - B // 2, 1), (0, B // 2, 0)] if halves else [(c0, min(chunk, B - c0), 0) for c0 in range(0, B, chunk)]
            xs_of = {c0: sub(xin, c0, nb) for c0, nb, _ in parts}
            for rec, (c0, nb, ln) in ([(r, p) for r in blocks for p in parts] if halves else [(r, p) for p in parts for r in blocks]):
                    xs = xs_of[c0]
                    nm, stride = rec['nm'], rec['stride']
                    a_, y_ = sub(rec['a'], c0, nb), sub(rec['y'], c0, nb)
                    # the projection shortcut of a stage's first block is independent of branch2a / 2b: on a side stream it runs beside them
                    # and branch2c (or the fused tail) joins it (GPP_BR1_LANE=0: serial, behind branch2a)
                    join_halves = lane_open and not halves       # first launch of a whole-batch stage behind a split one
                    side = br1_lane and rec['sc'] is not None and not halves and not join_halves
                    if side:
                        sc_ = sub(rec['sc'], c0, nb)
                        self._conv(plan, 'res{}_branch1'.format(nm), [xs], [sc_], 1, stride=stride, lane=1)
                    rec['block'] = f in fuse_block and f in (64, 128) and (rec['sc'] is None or fuse_block_proj) and xs.split     # (res2a reads the pooled map, which is float32)
                    b_or_a = sub(rec['b'], c0, nb) if rec['b'] is not None else a_       # (a stage with fused tails has no branch2b map: the descriptors borrow branch2a's -- neither is written)
                    if rec['block'] and rec['sc'] is None:
                        self._block(plan, nm, xs, a_, b_or_a, y_, xs, join=join_halves, lane=ln)
                    elif not rec['block']:
                        self._conv(plan, 'res{}_branch2a'.format(nm), [xs], [a_], 1, stride=stride, relu=True, lane=ln, join=join_halves)
                    if rec['sc'] is not None and not side:
                        sc_ = sub(rec['sc'], c0, nb)
                        self._conv(plan, 'res{}_branch1'.format(nm), [xs], [sc_], 1, stride=stride, lane=ln, join=join_halves and rec['block'])
                    elif rec['sc'] is None:
                        sc_ = xs
                    if join_halves:
                        lane_open = False
                    if rec['block'] and rec['sc'] is not None:           # a projection block: the shortcut map first (side lane or in line), then the whole block
                        self._block(plan, nm, xs, a_, b_or_a, y_, sc_, stride=stride, join=side, lane=ln)
                    if rec['block']:
                        pass
                    elif rec['b'] is None:
                        self._tail(plan, nm, a_, y_, sc_, join=side, lane=ln)
                    else:
                        b_ = sub(rec['b'], c0, nb)
                        self._conv(plan, 'res{}_branch2b'.format(nm), [a_], [b_], 3, pad=(1, 1), relu=True, lane=ln)
                        self._conv(plan, 'res{}_branch2c'.format(nm), [b_], [y_], 1, relu=True, residuals=[sc_], join=side, lane=ln)
                    xs_of[c0] = y_
            lane_open = lane_open or halves
            feats.append(x)
        _, C3, C4, C5 = feats
        plan.features = {'stem': plan.stem_out, 'C2': feats[0], 'C3': C3, 'C4': C4, 'C5': C5}

        # ---- FPN into one pyramid tensor (B, sum(H_l*W_l), 512)
        shapes = anchor_utils.pyramid_shapes((H, Wd))
        if (C3.H, C3.W) != shapes[0] or (C4.H, C4.W) != shapes[1] or (C5.H, C5.W) != shapes[2]:
            raise RuntimeError('backbone / pyramid shape mismatch: {} vs {}'.format([(C3.H, C3.W), (C4.H, C4.W), (C5.H, C5.W)], shapes))
        pix = [h * w for h, w in shapes]
        total = sum(pix)
        lvl_off = [sum(pix[:i]) for i in range(5)]
        plan.n_anchors = total * anchor_utils.NUM_BASE_ANCHORS

        # dtype='bf16x3': the maps between the FPN / head layers -- 80 % of the FLOPs, all of them matrix-pipe bound -- are stored
        # PRE-SPLIT ([32 bf16 hi | 32 bf16 lo] per 32 channels, gpp_conv_desc.x3_split): written that way by the producing layer's
        # epilogue, read by the consumers without the per-fragment split on the vector ALU (GPP_X3_SPLIT=0: plain float32 maps).
        # The backbone maps stay float32: its layers are bound by their tile fill, not by the matrix pipe.
        x3s = x3_level >= 1

        def pyramid(c, dtype=None):
            buf = torch.empty((B, total, c), dtype=dtype or dt, device=dev)
            plan.keep.append(buf)
            sp = x3s and dtype is None
            return buf, [C.FMap(buf, B, shapes[i][0], shapes[i][1], c, off=lvl_off[i] * c, bstride=total * c, split=sp,
                                half=self.dtype if sp else 'bf16x3') for i in range(5)]

        def smap(h, w, c):
            f = fmap(h, w, c)
            return f.mark_split() if x3s else f

        pyr, P = pyramid(512)
        T5 = smap(C5.H, C5.W, 512)
        # P5 and the P6 -> ReLU -> P7 chain are small launches (a few dozen tiles) independent of the C4 / C3 chain:
        # they run on the side streams underneath the C4_reduced / P4 launches (182 workgroups each: 74 CUs idle), joined by the
        # fused first tower layer (GPP_FPN_LANES=0: serial)
        fpn_lanes = head_lanes or os.environ.get('GPP_FPN_LANES', '1') != '0'      # measured +0.8 % on the f16x3 step
        l_p5, l_p6 = (1, 2) if fpn_lanes else (0, 0)
        self._conv(plan, 'C5_reduced', [C5], [T5], 1, join=lane_open)
        self._conv(plan, 'P5', [T5], [P[2]], 3, pad=(1, 1), lane=l_p5)
        self._conv(plan, 'P6', [C5], [P[3]], 3, stride=2, pad=(C.same_pad(C5.H, 3, 2)[1], C.same_pad(C5.W, 3, 2)[1]), lane=l_p6)
        R6 = smap(shapes[3][0], shapes[3][1], 512)
        relu_d = ReluDesc(pyr.data_ptr() + P[3].off * self.esz, R6.buf.data_ptr(), P[3].bstride, R6.bstride,
                          pix[3] * 512, C.gpp_dtype(self.dtype) if x3s else C.gpp_storage_dtype(self.dtype), B)
        plan.add(OP_RELU, relu_d, 'C6_relu', lane=l_p6)
        plan.touch(relu_d, [Plan.span(P[3])], [Plan.span(R6)])
        plan.relu_io = (P[3], R6)
        self._conv(plan, 'P7', [R6], [P[4]], 3, stride=2,
                   pad=(C.same_pad(shapes[3][0], 3, 2)[1], C.same_pad(shapes[3][1], 3, 2)[1]), lane=l_p6)
        T4 = smap(C4.H, C4.W, 512)
        self._conv(plan, 'C4_reduced', [C4], [T4], 1, residuals=[T5])          # + UpsampleLike(P5, C4), fused
        # P4 (182 workgroups of 192 x 256: 71 % of the CUs) only feeds the towers: behind P5 on its side stream (re-forked: it needs T4), it
        # runs beside C3_reduced / P3 (GPP_P4_LANE=0: serial)
        p4_lane = 1 if (fpn_lanes and os.environ.get('GPP_P4_LANE', '1') != '0') else 0
        self._conv(plan, 'P4', [T4], [P[1]], 3, pad=(1, 1), lane=p4_lane, sync=bool(p4_lane))
        T3 = smap(C3.H, C3.W, 512)
        self._conv(plan, 'C3_reduced', [C3], [T3], 1, residuals=[T4])          # + UpsampleLike(P4, C3), fused
        self._conv(plan, 'P3', [T3], [P[0]], 3, pad=(1, 1))

        plan.features.update({'P{}'.format(i + 3): P[i] for i in range(5)})

        # ---- heads: every layer is one grouped launch over the five levels
        # layer 0 of the three towers shares its input: one fused launch (C_out = 896) into a wide
        # tensor; layers 1..3 read their channel slice of it (in_pitch > C_in)
        wide, wide_maps = pyramid(896)
        def slice_of(maps, c0, c):
            return [C.FMap(m.buf, B, m.H, m.W, c, off=m.off + c0, bstride=m.bstride, pitch=m.pitch, split=m.split, half=m.half) for m in maps]

        # (measured and rejected: the half-empty fourth 256-column tile of this 896-wide layer as its own 128-column launch
        # on a side stream -- the two launches do not pack into each other's partial rounds, no gain)
        self._conv(plan, 'pyramid_towers_0', P, wide_maps, 3, pad=(1, 1), relu=True, join=True)

        def tower(prefix, width, src, tag=0, lane=0):
            for i in range(1, 4):
                _, dst = pyramid(width)
                name = '{}_{}'.format(prefix, i)
                self._conv(plan, name, src, dst, 3, pad=(1, 1), relu=True, tag=tag, lane=lane)
                src = dst
            return src

        # the three towers are independent chains: optionally (GPP_HEAD_LANES=1) the two small ones run on side
        # streams, forked after the fused first layer and joined by the decode, so that their launches fill the
        # ramp-up / tail phases of the big regression-tower kernels
        l_dim, l_cls = (1, 2) if head_lanes else (0, 0)
        # Launch order (default, GPP_DECODE_OVERLAP=1): classification tower, regression tower, dimension tower.
        # The detection selection (threshold + sort + greedy NMS: one workgroup per image, latency-bound, 8 of the
        # 256 CUs) only needs the classification logits and the corner regressions, so it runs on a side stream
        # underneath the dimension tower; the full decode of the <= 100 survivors joins when every head is done.
        overlap = os.environ.get('GPP_DECODE_OVERLAP', '1') != '0' and not head_lanes and not self.osf
        plan.decode_overlap = overlap
        plan.side_lanes = {'fpn': fpn_lanes, 'branch1': br1_lane, 'p4': bool(p4_lane), 'half_batch_stages': sorted(half_stages)}

        def dim_tower():
            dim_t = tower('pyramid_regression_dim', 128, slice_of(wide_maps, 768, 128), lane=l_dim)
            plan.regression_dim, dim_o = pyramid(36, torch.float32)
            self._conv(plan, 'pyramid_regression_dim', dim_t, dim_o, 3, pad=(1, 1), out_f32=True, lane=l_dim)

        # GPP_CLS_LANE (default: on for B <= 2): the classification tower (+ the candidate pass behind it) on side lane 2 BESIDE the regression
        # tower instead of in front of it.  At batch 1 a tower launch fields 0.9 - 1.9 rounds of workgroups of one wavefront per SIMD: two
        # independent chains in flight fill the other half of every SIMD (measured: profiles/r5/b1_latency.json, field `plan_variants`).  `pyramid_regression_ops`
        # joins the lane, so the selection that follows sees the candidate keys; at batch 8 every launch fills the chip on its own (off).
        cls_lane = 2 if (overlap and os.environ.get('GPP_CLS_LANE', '1' if B <= 2 else '0') != '0') else 0
        plan.side_lanes['cls_tower'] = bool(cls_lane)

        def cls_tower():
            cls_t = tower('pyramid_classification', 256, slice_of(wide_maps, 512, 256), lane=l_cls or cls_lane)
            plan.cls_logits, cls_o = pyramid(96, torch.float32)
            self._conv(plan, 'pyramid_classification', cls_t, cls_o, 3, pad=(1, 1), out_f32=True, lane=l_cls or cls_lane)

        def reg_tower():
            reg_t = tower('pyramid_regression', 512, slice_of(wide_maps, 0, 512), tag=1)
            plan.regression, reg_o = pyramid(144, torch.float32)
            self._conv(plan, 'pyramid_regression_ops', reg_t, reg_o, 3, pad=(1, 1), out_f32=True, join=bool(cls_lane))

        detect_at = {}
        if overlap:
            cls_tower()
            detect_at['candidates'] = len(plan.ops)
            reg_tower()
            detect_at['select'] = len(plan.ops)
            dim_tower()
        else:
            dim_tower()
            cls_tower()
            reg_tower()

        # ---- decode + NMS (RegressBoxes, RegressDims, FilterDetections)
        D = MAX_DETECTIONS
        f32, i32 = torch.float32, torch.int32
        plan.boxes = torch.empty((B, D, 12), dtype=f32, device=dev)
        plan.dimensions = torch.empty((B, D, 3), dtype=f32, device=dev)
        plan.scores = torch.empty((B, D), dtype=f32, device=dev)
        plan.labels = torch.empty((B, D), dtype=i32, device=dev)
        plan.orientations = torch.empty((B, D), dtype=i32, device=dev)
        plan.anchor_index = torch.empty((B, D), dtype=i32, device=dev)
        plan.counts = torch.zeros((B,), dtype=i32, device=dev)
        need = hip.c_size_t(0)
        if self.osf and B > 16:
            raise ValueError('orientation_specific_filter=True handles at most 16 images per batch')
        size_fn = hip.lib().gpp_detect_osf_workspace_bytes if self.osf else hip.lib().gpp_detect_workspace_bytes
        hip.check(size_fn(B, plan.n_anchors, need), 'gpp_detect_workspace_bytes')
        plan.detect_ws = torch.empty((int(need.value),), dtype=torch.uint8, device=dev)
        anchors = self._anchor_table((H, Wd))
        dd = DetectDesc(plan.cls_logits.data_ptr(), plan.regression.data_ptr(), plan.regression_dim.data_ptr(),
                        anchors.data_ptr(), plan.boxes.data_ptr(), plan.dimensions.data_ptr(), plan.scores.data_ptr(),
                        plan.labels.data_ptr(), plan.orientations.data_ptr(), plan.anchor_index.data_ptr(),
                        plan.counts.data_ptr(), plan.detect_ws.data_ptr(), plan.detect_ws.numel(), plan.n_anchors,
                        B, anchor_utils.NUM_BASE_ANCHORS, 1, D, SCORE_THRESHOLD, NMS_THRESHOLD if self.nms else 2.0)
        # what the decode stages touch (one shared descriptor: logged per op kind)
        sp = Plan.span_of
        det_out = [sp(t) for t in (plan.boxes, plan.dimensions, plan.scores, plan.labels, plan.orientations, plan.anchor_index)]
        heads_in = [sp(plan.cls_logits), sp(plan.regression), sp(plan.regression_dim)]
        plan.touch(dd, [sp(plan.cls_logits)], [sp(plan.detect_ws), sp(plan.counts)], kind=OP_DETECT_CANDIDATES)
        plan.touch(dd, [sp(plan.detect_ws), sp(plan.regression)], [sp(plan.detect_ws)], kind=OP_DETECT_SELECT)
        plan.touch(dd, heads_in + [sp(plan.detect_ws)], det_out, kind=OP_DETECT_EMIT)
        for whole in (OP_DETECT, OP_DETECT_OSF):
            plan.touch(dd, heads_in, det_out + [sp(plan.detect_ws), sp(plan.counts)], kind=whole)
        if overlap:
            # spliced in where their inputs are complete (the descriptors need the buffers allocated above)
            at = detect_at['select']
            plan.ops.insert(at, (OP_DETECT_SELECT, 0, dd, 'filtered_detections/select', 0.0))
            plan.lanes.insert(at, (1 << 8) | OP_SYNC)
            at = detect_at['candidates']
            plan.ops.insert(at, (OP_DETECT_CANDIDATES, 0, dd, 'filtered_detections/candidates', 0.0))
            plan.lanes.insert(at, (cls_lane or 1) << 8)             # behind the classification tower: on its lane when it has one
            plan.add(OP_DETECT_EMIT, dd, 'filtered_detections', join=True)
        else:
            plan.add(OP_DETECT_OSF if self.osf else OP_DETECT, dd, 'filtered_detections', join=True)

        # ---- ground-plane polling (FitRoadPlanes)
        plan.keypoints = torch.empty((B, D, 4, 3), dtype=f32, device=dev)
        plan.keyplanes = torch.empty((B, D, 1, 4), dtype=f32, device=dev)
        plan.residuals = torch.empty((B, D), dtype=f32, device=dev)
        plan.best_index = torch.empty((B, D), dtype=i32, device=dev)
        hip.check(hip.lib().gpp_poll_workspace_bytes(B, n_planes, int(planes_batched), need), 'gpp_poll_workspace_bytes')
        plan.poll_ws = torch.empty((max(int(need.value), 16),), dtype=torch.uint8, device=dev)
        pd = PollDesc(plan.boxes.data_ptr(), plan.dimensions.data_ptr(), plan.orientations.data_ptr(), plan.P_inv.data_ptr(),
                      plan.planes.data_ptr(), plan.keypoints.data_ptr(), plan.keyplanes.data_ptr(), plan.residuals.data_ptr(),
                      plan.best_index.data_ptr(), plan.poll_ws.data_ptr(), plan.poll_ws.numel(), B, D, n_planes,
                      int(planes_batched), POLL_THRESHOLD, 0)
        plan.add(OP_POLL, pd, 'fit_road_planes', tag=2, flops=162.0 * B * D * n_planes)      # tag 2: bench.py times it live too
        plan.touch(pd, [Plan.span_of(t) for t in (plan.boxes, plan.dimensions, plan.orientations, plan.P_inv, plan.planes)],
                   [Plan.span_of(t) for t in (plan.keypoints, plan.keyplanes, plan.residuals, plan.best_index, plan.poll_ws)])
        plan.workspaces = {lane: torch.empty((max(need_, 16),), dtype=torch.uint8, device=dev) for lane, need_ in plan.ws_need.items()}
        for d, lane in plan.conv_descs:
            d.partial = plan.workspaces[lane].data_ptr()
            d.partial_bytes = plan.workspaces[lane].numel()
        plan.finalize()
        plan.tagged = [name for _, tag, _, name, _ in plan.ops if tag]
        if os.environ.get('GPP_AUTOTUNE', '1') != '0':
            self._autotune(plan)
        return plan

    # ------------------------------------------------------------------ per-layer tile selection
    def _tune_cache_path(self):
        return os.environ.get('GPP_TUNE_CACHE')

    def _tune_config(self):
        """ what a cached tile choice is valid for besides (backbone, type, layer, batch, image size): the build of the library (tile codes
        come and go with it: gpp_version() carries a hash of the kernel sources) and the plan options that change which maps are
        pre-split or fused -- a tile timed on a float32 map may not even exist for the pre-split form of the same layer """
        ver = hip.lib().gpp_version().decode().split('src:')[-1]
        return 'v2;{};x3split={};fuse={};plan={}{}'.format(ver, os.environ.get('GPP_X3_SPLIT', '2'), os.environ.get('GPP_FUSE_TAIL', '64,128') + '/' + os.environ.get('GPP_FUSE_BLOCK', '64,128'), self.plan_mode,
                                                           C.latency_split_config() if self.plan_mode == 'latency' else '')

    def _load_tune_cache(self):
        path = self._tune_cache_path()
        if path and os.path.exists(path):
            with open(path) as f:
                for key, val in json.load(f).items():
                    parts = key.split('|')
                    if len(parts) != 7:              # (files of rounds 1-3: no library hash in the key -- ignored, the layers are timed again)
                        continue
                    cfg, bb, dt, name, b, h, w = parts
                    if cfg == self._tune_config() and bb == self.backbone_name and dt == self.dtype:
                        self._tuned[(name, int(b), int(h), int(w))] = (int(val[0]), float(val[-1]))

    def _save_tune_cache(self):
        """ Rank 0 only, through a temporary file + os.replace: readers never see a torn file, ranks never race. """
        path = self._tune_cache_path()
        if not path or int(os.environ.get('RANK', '0')) != 0:
            return
        data = {}
        if os.path.exists(path):
            try:
                with open(path) as f:
                    data = json.load(f)
            except ValueError:
                data = {}
        for (name, b, h, w), val in self._tuned.items():
            data['|'.join([self._tune_config(), self.backbone_name, self.dtype, name, str(b), str(h), str(w)])] = list(val)
        tmp = '{}.tmp.{}'.format(path, os.getpid())
        with open(tmp, 'w') as f:
            json.dump(data, f, indent=0, sort_keys=True)
        os.replace(tmp, path)

    def _autotune(self, plan):
        """ Choose the block tile of every conv layer of this plan by timing the candidates on the device
        (gpp_conv2d_autotune), layer by layer on realistic activations (a noise frame pushed through the layers
        before).  Whether a layer's tile grid fills the 256 CUs in 1.07 or 0.95 rounds decides its time by up to 1.5x
        and is cheap to measure.  The tile NEVER changes a result (same K order per output element,
        test_every_tile_gives_identical_results), so ranks and plans may choose differently without any effect on the
        outputs; split-K, which would, is not tuned (gpp_conv2d_split_rule).  Choices are remembered per (layer, batch,
        image size) and can be persisted with GPP_TUNE_CACHE=<file.json>.  GPP_AUTOTUNE=0 keeps the library heuristic. """
        torch = self.torch
        B, H, Wd = plan.shape[:3]
        plan.images.uniform_(-120.0, 130.0)
        best = ctypes.c_float(0.0)
        fresh = False
        plan.tuning = {}
        # GPP_TUNE_RANDOM=<seed> (tests / tools/first_run_stress.py): every layer takes a RANDOM tile among those the library
        # accepts for it instead of the fastest one -- the outputs may not change by a bit, whatever the draw
        rnd = None
        if os.environ.get('GPP_TUNE_RANDOM'):
            import random
            rnd = random.Random(int(os.environ['GPP_TUNE_RANDOM']) * 1000003 + int(os.environ.get('RANK', '0')))
        for index, (kind, _, desc, name, flops) in enumerate(plan.ops):
            if kind in DETECT_OPS or kind == OP_POLL:
                continue
            self.run_op(plan, index)
            if rnd is not None and kind in (OP_TAIL, OP_CONV):
                if kind == OP_CONV:
                    tiles, count = (ctypes.c_int * 32)(), ctypes.c_int(0)
                    hip.check(hip.lib().gpp_conv2d_tile_candidates(ctypes.byref(desc), tiles, 32, ctypes.byref(count)), 'gpp_conv2d_tile_candidates')
                    ok = []
                    for tile in tiles[:count.value]:         # a candidate the launcher refuses for this shape is skipped, as the autotuner does
                        desc.tile_hint = tile
                        if hip.lib().gpp_plan_run(ctypes.byref(plan.array, index * ctypes.sizeof(PlanOp)), 1, hip.stream_ptr(), None, 0) == 0:
                            ok.append(tile)
                    desc.tile_hint = rnd.choice(ok)
                    plan.tuning[name] = (int(desc.tile_hint), 0.0)
                else:
                    desc.tile_rows = rnd.choice((64, 96, 128, 160) if self.dtype in C.X3_TYPES else (96, 128, 160))
                    plan.tuning[name] = (int(desc.tile_rows), 0.0)
                self.run_op(plan, index)
                continue
            if kind == OP_TAIL:
                key = (name, plan.op_batch.get(id(desc), B), H, Wd)          # (a half-batch launch is tuned as what it is)
                if key not in self._tuned:
                    times = {}
                    for rows in ((64, 96, 128, 160) if (self.dtype in C.X3_TYPES and os.environ.get('GPP_TAIL64', '1') != '0') else (96, 128, 160)):
                        desc.tile_rows = rows
                        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        self.run_op(plan, index)
                        e0.record()
                        for _ in range(16):
                            self.run_op(plan, index)
                        e1.record()
                        e1.synchronize()
                        times[rows] = e0.elapsed_time(e1) * 1000.0 / 16
                    rows = min(times, key=times.get)
                    self._tuned[key] = (rows, round(times[rows], 2))
                    fresh = True
                desc.tile_rows = self._tuned[key][0]
                plan.tuning[name] = self._tuned[key]
                plan.tuning_parts.setdefault(name, []).append(self._tuned[key])
                continue
            if kind != OP_CONV:
                continue
            key = (name, plan.op_batch.get(id(desc), B), H, Wd)
            if key in self._tuned and not self._tile_is_listed(desc, self._tuned[key][0]):
                del self._tuned[key]                 # a remembered tile this build does not offer for this layer: time the layer again
            if key not in self._tuned:
                # launches per candidate and repetition (the library times two repetitions and keeps the faster).  The float32 path's launches are
                # 3 x as long as the x3 types': fewer of them.  Round 4: the x3 types used the float32 counts (2 / 4) -- a big layer's choice then rested
                # on four launches per candidate, and one tuning run in ~60 picked a tile that cost the whole run 5 %
                slow = self.dtype == 'f32'
                iters = (2 if slow else 6) if flops > 5e10 else (4 if slow else 16)
                hip.check(hip.lib().gpp_conv2d_autotune(ctypes.byref(desc), iters, hip.stream_ptr(), ctypes.byref(best)),
                          'gpp_conv2d_autotune')
                self._tuned[key] = (int(desc.tile_hint), round(float(best.value), 2))
                fresh = True
            desc.tile_hint = self._tuned[key][0]
            plan.tuning[name] = self._tuned[key]
            plan.tuning_parts.setdefault(name, []).append(self._tuned[key])
        torch.cuda.synchronize()
        if fresh:
            self._save_tune_cache()

    @staticmethod
    def _tile_is_listed(desc, tile):
        tiles, count = (ctypes.c_int * 64)(), ctypes.c_int(0)
        hip.check(hip.lib().gpp_conv2d_tile_candidates(ctypes.byref(desc), tiles, 64, ctypes.byref(count)), 'gpp_conv2d_tile_candidates')
        return int(tile) in list(tiles[:min(count.value, 64)])

    def x3_range_events(self, reset=False):
        """ dtype='f16x3': how many 8-channel groups the epilogues of THIS model's plans have stored with a value outside the half range (a finite
        activation beyond +-65504, which is clamped, or an inf / NaN, which stays one) since the counters were last reset.  Every plan counts
        into a slot of its own (Plan.range_slot): other models on the device, other plans and their resets do not show here, and a reset here
        touches nothing of theirs.  Zero = the type's range altered nothing.  Synchronises. """
        if self.dtype != 'f16x3':
            return 0
        self.torch.cuda.synchronize()
        total = 0
        for plan in self._plans.values():
            total += int(plan.range_slot.item())
            if reset:
                plan.range_slot.zero_()
                plan.range_seen = 0
        if reset:
            self.torch.cuda.synchronize()
        return total

    def note_range(self, plan, count):
        """ count = the plan's counter as a fetched result saw it: True when events happened since the plan's previous fetch """
        if count == plan.range_seen:
            return False
        plan.range_seen = count
        return True

    def plan_for(self, B, H, Wd, n_planes, planes_batched):
        key = (int(B), int(H), int(Wd), int(n_planes), bool(planes_batched))
        if key not in self._plans:
            self._plans[key] = self._build(*key)
        return self._plans[key]

    # ------------------------------------------------------------------ execution
    def run_op(self, plan, index):
        """ Enqueue ONE op of the plan (per-layer tests). """
        hip.check(hip.lib().gpp_plan_run(ctypes.byref(plan.array, index * ctypes.sizeof(PlanOp)), 1, hip.stream_ptr(), None, 0), 'gpp_plan_run')

    def run_plan(self, plan, events=None):
        """ Enqueue the whole forward on the current stream (asynchronous). """
        if events is None and getattr(plan, 'graph', None) is not None:
            plan.graph.replay()
            return
        if events is not None:
            arr = (ctypes.c_void_p * len(events))(*events)
            rc = hip.lib().gpp_plan_run(plan.array, len(plan.ops), hip.stream_ptr(), arr, len(events))
        else:
            rc = hip.lib().gpp_plan_run(plan.array, len(plan.ops), hip.stream_ptr(), None, 0)
        hip.check(rc, 'gpp_plan_run')

    def capture(self, plan):
        """ Record the plan into a HIP graph (via torch's stream capture); later run_plan(plan) calls replay
        it with one launch.  Measured on MI355X: no gain (B = 1: 2.00 -> 1.98 ms, B = 8: 5.44 -> 5.43 ms) -- the plan
        is GPU-bound, kernel time is 99 % of the step -- so it is optional and off by default. """
        torch = self.torch
        self.run_plan(plan)                      # warm-up outside the capture (one-time kernel attribute calls)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            rc = hip.lib().gpp_plan_run(plan.array, len(plan.ops), hip.stream_ptr(), None, 0)
        hip.check(rc, 'gpp_plan_run (capture)')
        plan.graph = graph
        return graph

    def outputs(self, plan):
        """ the 8 device tensors in the reference's output order (retinanet.py:418-419) """
        return [plan.boxes, plan.dimensions, plan.scores, plan.labels, plan.orientations,
                plan.keypoints, plan.keyplanes, plan.residuals]

    def predict_on_batch(self, inputs):
        """ inputs = [images (B, H, W, 3) float32 BGR mean-subtracted, P_inv (B, 4, 3), planes (B, N, 4)]
        (NumPy, as bin/run_network.py:105 builds them, or torch tensors already on the device).
        Returns the list of 8 writable NumPy arrays of the reference:
        boxes (B,100,12) f32, dimensions (B,100,3) f32, scores (B,100) f32, labels (B,100) i32,
        orientations (B,100) i32, keypoints (B,100,4,3) f32, keyplanes (B,100,1,4) f32, residuals (B,100) f32. """
        plan = self.stage_inputs(inputs)
        self.run_plan(plan)
        return self.fetch(plan)

    def fetch(self, plan, packed=None):
        """ the 8 result arrays of the plan's last run as writable NumPy arrays (synchronises).  packed (default; GPP_FETCH=separate for
        the other form): ONE launch packs them into a (B, 100, 35) float32 tensor (gpp_pack_detections; labels / orientations are small
        integers: exact) and ONE copy brings it to the host, instead of eight blocking copies of 0.4 - 4.8 KB per image -- the same bytes
        (tests/test_network_gpu.py), 8 -> 1 host round trips inside the bracket the reference times (bin/run_network.py:108-111). """
        if packed is None:
            packed = os.environ.get('GPP_FETCH', 'packed') != 'separate'
        watch = self.watches_range()
        if packed:
            outs, count = self.unpack_with_range(self.pack_with_range(plan).cpu().numpy(), plan.shape[0])
        else:
            outs = [t.cpu().numpy() for t in self.outputs(plan)]
            count = int(self.range_snapshot(plan).cpu().view(self.torch.int64).item()) if watch else 0
        if watch and self.note_range(plan, count):
            return self._range_event([plan.images, plan.P_inv, plan.planes], 'predict_on_batch')
        return outs

    # ------------------------------------------------------------------ f16x3: the half range, watched
    def watches_range(self):
        return self.dtype == 'f16x3' and self.on_range_event != 'ignore'

    def range_snapshot(self, plan, dst=None):
        """ enqueue a copy of the plan's range-event counter (its value at this point of the current stream) into two float32 words
        of device memory (gpp_x3_range_snapshot_of); no synchronisation """
        if dst is None:
            dst = self.torch.empty((2,), dtype=self.torch.float32, device=self.device)
        hip.check(hip.lib().gpp_x3_range_snapshot_of(ctypes.c_void_p(plan.range_slot.data_ptr()), ctypes.c_void_p(dst.data_ptr()), hip.stream_ptr()),
                  'gpp_x3_range_snapshot_of')
        return dst

    def pack_with_range(self, plan, out=None):
        """ the results of the plan's last run as ONE flat float32 device buffer: B x 100 x 35 packed detections (gpp_pack_detections)
        followed by the 8 bytes of the range-event counter as the stream saw it behind them -- one copy brings both to the host """
        from ..utils import distributed as D
        torch = self.torch
        outs = self.outputs(plan)
        B, Dn = int(outs[0].shape[0]), int(outs[0].shape[1])
        n = B * Dn * D.PACK_WIDTH
        if out is None:
            out = torch.empty((n + 2,), dtype=torch.float32, device=self.device)
        hip.check(hip.lib().gpp_pack_detections(*([hip.ptr(o) for o in outs] + [B, Dn, ctypes.c_void_p(out.data_ptr()), hip.stream_ptr()])),
                  'gpp_pack_detections')
        if self.watches_range():
            self.range_snapshot(plan, out[n:])
        return out

    @staticmethod
    def unpack_with_range(flat, B):
        """ host side of pack_with_range: (the 8 NumPy result arrays, the counter value) """
        from ..utils import distributed as D
        flat = np.ascontiguousarray(flat)
        n = flat.size - 2
        return D.unpack_outputs(flat[:n].reshape(B, n // (B * D.PACK_WIDTH), D.PACK_WIDTH)), int(flat[n:].view(np.uint64)[0])

    def _range_event(self, device_inputs, what):
        """ an activation of the call just fetched left the half range: its result is not the reference's.  device_inputs = the call's
        [images, P_inv, planes] still in HBM. """
        self.range_fallbacks += 1
        if self.on_range_event == 'raise':
            raise hip.GppError('{}: an activation left the IEEE-half range of dtype=\'f16x3\' (finite beyond +-65504, inf or NaN; '
                               'gpp_x3_range_events): the result would not be the reference\'s -- load the model with dtype=\'f32\' '
                               'or on_range_event=\'f32\''.format(what))
        self.prepare_fallback()
        if what == 'predict_on_frames':          # device_inputs = [frames uint8, P_inv, planes]: preprocessing included
            return self._twin.predict_on_frames(*device_inputs)[0]
        plan = self._twin.stage_inputs(device_inputs)
        self._twin.run_plan(plan)
        return self._twin.fetch(plan)

    def prepare_fallback(self, B=None, H=None, Wd=None, n_planes=None, planes_batched=True):
        """ on_range_event='f32': build the float32 twin NOW -- its weights upload (a second copy of the weights in HBM) and, when a shape is
        given, its plan for that shape (buffers + tile tuning: seconds) -- instead of inside the first call whose activations leave the half
        range, where it would stall a latency-critical predict_on_batch / FramePipeline iteration (and every other rank of a sharded call).
        Without it the twin is built lazily, at the first event.  The host copy of the weights is dropped once the twin exists. """
        if self.dtype != 'f16x3' or self.on_range_event != 'f32':
            return None
        if self._twin is None:
            self._twin = RetinaNet3D(self._weights, backbone_name=self.backbone_name, dtype='f32', nms=self.nms,
                                     class_specific_filter=self.class_specific_filter, orientation_specific_filter=self.osf,
                                     name=self.name + '-f32-twin', plan=self.plan_mode)
            self._weights = None
        if B is not None:
            self._twin.plan_for(B, H, Wd, n_planes, planes_batched)
        return self._twin

    def stage_inputs(self, inputs):
        """ Copy [images, P_inv, planes] into the plan's device buffers; returns the plan. """
        torch = self.torch
        if not isinstance(inputs, (list, tuple)) or len(inputs) != 3:
            raise ValueError('predict_on_batch expects [images, P_inv, planes]')
        images, P_inv, planes = inputs
        shp = tuple(images.shape)
        if len(shp) != 4 or shp[3] != 3:
            raise ValueError('images must be (B, H, W, 3), got {}'.format(shp))
        if tuple(P_inv.shape) != (shp[0], 4, 3):
            raise ValueError('P_inv must be (B, 4, 3), got {}'.format(tuple(P_inv.shape)))
        pshape = tuple(planes.shape)
        batched = len(pshape) == 3
        if not ((batched and pshape[0] == shp[0] and pshape[2] == 4) or (len(pshape) == 2 and pshape[1] == 4)) or pshape[-2] < 1:
            raise ValueError('planes must be (B, N, 4) or (N, 4), got {}'.format(pshape))
        plan = self.plan_for(shp[0], shp[1], shp[2], pshape[-2], batched)

        def put(dst, src):
            if isinstance(src, torch.Tensor):
                dst.copy_(src.to(dtype=dst.dtype), non_blocking=True)
            else:
                dst.copy_(torch.as_tensor(np.ascontiguousarray(src, dtype=np.float32)), non_blocking=True)

        if not isinstance(images, torch.Tensor) and os.environ.get('GPP_UPLOAD', 'pageable') == 'pinned':
            # GPP_UPLOAD=pinned: host frames go through a page-locked staging buffer of the plan (one memcpy on the host, then a DMA).
            # Measured at batch 1 (tools/b1_latency.py --host-variants, profiles/r5/b1_latency.json): 0.37 ms for the 6.4 MB float32 frame
            # against 0.16 - 0.17 ms for the runtime's own staged copy from pageable memory -- the host memcpy costs more than it saves,
            # so the plain copy is the default
            if getattr(plan, 'host_images', None) is None:
                plan.host_images = torch.empty(tuple(plan.images.shape), dtype=torch.float32, pin_memory=True)
                plan.host_images_free = torch.cuda.Event()
            else:
                plan.host_images_free.synchronize()          # the previous upload out of this buffer has left it
            np.copyto(plan.host_images.numpy(), images, casting='same_kind')
            plan.images.copy_(plan.host_images, non_blocking=True)
            plan.host_images_free.record()
        else:
            put(plan.images, images)
        put(plan.P_inv, P_inv)
        put(plan.planes, planes)
        return plan

    # ------------------------------------------------------------------ raw frames (GPU preprocessing)
    def stage_frames(self, frames_u8, P_inv, planes, min_side=800, max_side=1333):
        """ frames_u8 (B, H, W, 3) uint8 BGR as utils.image.read_image_bgr returns them.  Uploads the raw
        bytes and runs mean subtraction + bilinear resize on the device (csrc/preprocess.hip), i.e. what
        bin/run_network.py:95-99 does on the host.  Returns (plan, scale). """
        from ..utils import image as image_utils
        torch = self.torch
        frames = frames_u8 if isinstance(frames_u8, torch.Tensor) else torch.as_tensor(np.ascontiguousarray(frames_u8))
        if frames.dtype != torch.uint8 or frames.dim() != 4 or frames.shape[3] != 3:
            raise ValueError('frames must be (B, H, W, 3) uint8, got {} {}'.format(tuple(frames.shape), frames.dtype))
        B, H, Wd = int(frames.shape[0]), int(frames.shape[1]), int(frames.shape[2])
        scale = image_utils.compute_resize_scale((H, Wd, 3), min_side, max_side)
        Ho, Wo = int(np.rint(H * scale)), int(np.rint(Wd * scale))
        key = (H, Wd, Ho, Wo)
        if not hasattr(self, '_taps'):
            self._taps = {}
        if key not in self._taps:
            y0, y1, wy = image_utils._axis_taps(Ho, H, scale)
            x0, x1, wx = image_utils._axis_taps(Wo, Wd, scale)
            dev = self.device
            self._taps[key] = [torch.as_tensor(a.astype(np.int32)).to(dev) for a in (y0, y1)] + [torch.as_tensor(wy).to(dev)] + \
                              [torch.as_tensor(a.astype(np.int32)).to(dev) for a in (x0, x1)] + [torch.as_tensor(wx).to(dev)]
        y0, y1, wy, x0, x1, wx = self._taps[key]
        pshape = tuple(planes.shape)
        plan = self.plan_for(B, Ho, Wo, pshape[-2], len(pshape) == 3)
        frames_d = frames.to(self.device, non_blocking=True).contiguous()
        m = image_utils.IMAGENET_MEAN_BGR
        hip.check(hip.lib().gpp_preprocess_u8_bgr(hip.ptr(frames_d), hip.ptr(plan.images), hip.ptr(y0), hip.ptr(y1), hip.ptr(wy),
                                                  hip.ptr(x0), hip.ptr(x1), hip.ptr(wx), B, H, Wd, Ho, Wo,
                                                  float(m[0]), float(m[1]), float(m[2]), hip.stream_ptr()), 'gpp_preprocess_u8_bgr')

        def put(dst, src):
            src_t = src if isinstance(src, torch.Tensor) else torch.as_tensor(np.ascontiguousarray(src, dtype=np.float32))
            dst.copy_(src_t.to(dtype=dst.dtype), non_blocking=True)

        put(plan.P_inv, P_inv)
        put(plan.planes, planes)
        plan.keep_frames = frames_d
        return plan, scale

    def predict_on_frames(self, frames_u8, P_inv, planes):
        """ predict_on_batch for raw uint8 BGR frames; returns (the 8 output arrays, scale).  P_inv must
        have been computed for that scale (utils.image.compute_resize_scale). """
        plan, scale = self.stage_frames(frames_u8, P_inv, planes)
        self.run_plan(plan)
        return self.fetch(plan), scale

    # Keras-style conveniences used by the reference's scripts
    def predict(self, inputs, batch_size=None, verbose=0):
        return self.predict_on_batch(inputs)

    def summary(self):
        print('{}: {} + FPN + heads, {} storage, {} conv launches'.format(
            self.name, self.backbone_name, self.dtype, len(self.conv_w)))
