"""
dtype='f16x3' must never return a clamped activation as a plausible wrong answer (round-4 review, weak #3).

The IEEE-half halves of the headline type end at +-65504; an epilogue that has to clamp a stored activation counts it
(gpp_x3_range_events).  The model READS that counter with the results every synchronous call fetches anyway (model.fetch,
FramePipeline._collect: the 8 bytes ride behind the packed detections, no extra synchronisation) and reacts (`on_range_event`):
'f32' (default) runs the call again on a float32 twin and returns the reference-precision result, 'raise' refuses, 'ignore' is the old
behaviour.

The weights that provoke it compute the SAME function as the seeded ones: bn2a_branch2a's (gamma, beta) x 2^17 and res2a_branch2b's
kernel x 2^-17 -- a ReLU commutes with a positive scale and powers of two are exact, so at float32 every tensor behind branch2b is
bit-identical to the unscaled network's, while the map between the two layers holds values beyond 65504.
"""
import numpy as np
import pytest

import helpers
from oracle import decode_np, net_torch
from keras_retinanet_3D import models
from keras_retinanet_3D.backend import hip
from keras_retinanet_3D.models import weights as W
from keras_retinanet_3D.utils import ledger, synthetic

pytestmark = pytest.mark.gpu

B, H, Wd = 2, 96, 160
SCALE = np.float32(2.0 ** 17)


def scaled_weights():
    w = dict(W.synthetic_weights('resnet50', 1234))
    w['bn2a_branch2a/gamma'] = w['bn2a_branch2a/gamma'] * SCALE
    w['bn2a_branch2a/beta'] = w['bn2a_branch2a/beta'] * SCALE
    w['res2a_branch2b/kernel'] = w['res2a_branch2b/kernel'] / SCALE
    return w


def inputs():
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, size=(B, H, Wd, 3)).astype(np.float32) - np.array([103.939, 116.779, 123.68], np.float32)
    planes = synthetic.load_plane_database('100').astype(np.float32)
    _, P_inv = synthetic.synthetic_calibration()
    return [img, np.tile(P_inv[None].astype(np.float32), (B, 1, 1)), np.tile(planes[None], (B, 1, 1))]


def same(a, b):
    return all(helpers.bits_equal(x, y) if x.dtype.kind == 'f' else np.array_equal(x, y) for x, y in zip(a, b)) and len(a) == len(b) == 8


@pytest.fixture(scope='module')
def f32_of_the_base_weights():
    return models.load_model('synthetic:1234', backbone_name='resnet50', dtype='f32').predict_on_batch(inputs())


def test_a_call_whose_activations_leave_the_half_range_returns_the_float32_result(f32_of_the_base_weights, oracle_lib):
    w = scaled_weights()
    model = models.load_model(w, backbone_name='resnet50', dtype='f16x3')                    # on_range_event='f32' is the default
    assert model.on_range_event == 'f32' and model.range_fallbacks == 0
    out = model.predict_on_batch(inputs())
    assert model.range_fallbacks == 1
    want = models.load_model(w, backbone_name='resnet50', dtype='f32').predict_on_batch(inputs())
    assert same(out, want)                                    # byte for byte what dtype='f32' returns for these weights
    assert same(out, f32_of_the_base_weights)                 # ... which is what the unscaled network returns at float32 (exact scaling)
    assert int((out[2] > 0.05).sum()) > 0
    # and that is the float32 CPU oracle's answer: same detections, orientations, plane indices (as __graft_entry__.smoke() checks f32)
    from keras_retinanet_3D.utils import anchors as A
    f = net_torch.forward(W.synthetic_weights('resnet50', 1234), inputs()[0], 'resnet50', storage=None)
    det, aidx = decode_np.detect(f['classification_logits'], f['regression'], f['regression_dim'], A.anchors_for_image((H, Wd)))
    kp, kpl, res, idx = helpers.c_oracle_poll(oracle_lib, det[0], det[1], det[4], inputs()[1], inputs()[2])
    twin_plan = model._twin.plan_for(B, H, Wd, 100, True)
    led = ledger.parity_ledger(list(det) + [kp, kpl, res], aidx, idx, out, twin_plan.anchor_index.cpu().numpy(), twin_plan.best_index.cpu().numpy())
    assert led['set_differences'] == 0 and led['same_orientation'] == led['common'] and led['same_plane'] == led['common'] > 0, led
    assert led['max_keypoint_rel_dev'] <= 1e-4, led
    # every further call is watched too
    again = model.predict_on_batch(inputs())
    assert model.range_fallbacks == 2 and same(again, want)


def test_raise_and_ignore():
    w = scaled_weights()
    strict = models.load_model(w, backbone_name='resnet50', dtype='f16x3')
    strict.on_range_event = 'raise'
    with pytest.raises(hip.GppError, match='half range'):
        strict.predict_on_batch(inputs())
    loose = models.load_model(w, backbone_name='resnet50', dtype='f16x3')
    loose.on_range_event = 'ignore'
    loose.x3_range_events(reset=True)
    got = loose.predict_on_batch(inputs())                    # the clamped map goes through: a finite, wrong answer
    assert loose.x3_range_events() > 0 and loose.range_fallbacks == 0
    want = models.load_model(w, backbone_name='resnet50', dtype='f32').predict_on_batch(inputs())
    assert not same(got, want)


def test_sane_weights_never_take_the_fallback_and_the_packed_fetch_equals_the_eight_copies(f32_of_the_base_weights):
    model = models.load_model('synthetic:1234', backbone_name='resnet50', dtype='f16x3')
    out = model.predict_on_batch(inputs())
    plan = model.plan_for(B, H, Wd, 100, True)
    separate = model.fetch(plan, packed=False)
    assert model.range_fallbacks == 0 and model._twin is None
    assert same(out, separate)
    for a, b in zip(out, separate):
        assert a.dtype == b.dtype and a.shape == b.shape and a.flags.writeable
    assert int((out[2] > 0.05).sum()) == int((f32_of_the_base_weights[2] > 0.05).sum())       # (parity of the type: tests/test_fullsize_*.py)


def test_frame_pipeline_reruns_an_affected_batch_at_float32():
    from keras_retinanet_3D.utils import image
    from keras_retinanet_3D.utils.pipeline import FramePipeline
    w = scaled_weights()
    planes = synthetic.load_plane_database('100').astype(np.float32)
    scale = image.compute_resize_scale((375, 1242, 3))
    _, P_inv = synthetic.synthetic_calibration(scale)
    P_inv = np.tile(P_inv[None].astype(np.float32), (1, 1, 1))
    batches = [((np.random.default_rng(k).integers(0, 2, size=(1, 375, 1242, 3)) * 255).astype(np.uint8), P_inv, planes) for k in range(3)]
    ref = models.load_model(w, backbone_name='resnet50', dtype='f32')
    want = [ref.predict_on_frames(*b)[0] for b in batches]
    model = models.load_model(w, backbone_name='resnet50', dtype='f16x3')
    got = list(FramePipeline(model, depth=3).run(iter(batches)))
    assert model.range_fallbacks == 3
    for (outs, _), r in zip(got, want):
        assert same(outs, r)


def test_a_sharded_call_recomputes_its_shard_before_the_gather():
    """ utils.distributed.ShardedModel: the rank's shard is checked (and, after an event, recomputed at float32) before it goes on the wire """
    from keras_retinanet_3D.utils import distributed as D
    w = scaled_weights()
    model = models.load_model(w, backbone_name='resnet50', dtype='f16x3')
    got = D.ShardedModel(model).predict_on_batch(inputs())                # (no process group: one rank holds the whole batch)
    want = models.load_model(w, backbone_name='resnet50', dtype='f32').predict_on_batch(inputs())
    assert model.range_fallbacks == 1 and same(got, want)
    sane = models.load_model('synthetic:1234', backbone_name='resnet50', dtype='f16x3')
    a = D.ShardedModel(sane).predict_on_batch(inputs())
    b = sane.predict_on_batch(inputs())
    assert sane.range_fallbacks == 0 and same(a, b)


def test_two_models_on_two_streams_only_the_affected_one_falls_back(f32_of_the_base_weights):
    """ every plan counts its range events into a slot of its own (gpp_conv_desc.range_counter, Plan.range_slot): a model whose activations
    leave the half range beside a sane one -- interleaved calls on two streams of one device, resets of either -- makes exactly its own
    calls fall back; the sane model never does, and its counters never move """
    import torch
    bad = models.load_model(scaled_weights(), backbone_name='resnet50', dtype='f16x3')
    sane = models.load_model('synthetic:1234', backbone_name='resnet50', dtype='f16x3')
    want_bad = models.load_model(scaled_weights(), backbone_name='resnet50', dtype='f32').predict_on_batch(inputs())
    want_sane = sane.predict_on_batch(inputs())
    assert sane.range_fallbacks == 0
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    for k in range(3):
        with torch.cuda.stream(s1):
            plan_bad = bad.stage_inputs(inputs())
            bad.run_plan(plan_bad)
        with torch.cuda.stream(s2):
            plan_sane = sane.stage_inputs(inputs())
            sane.run_plan(plan_sane)
        with torch.cuda.stream(s2):
            got_sane = sane.fetch(plan_sane)
        with torch.cuda.stream(s1):
            got_bad = bad.fetch(plan_bad)
        assert same(got_sane, want_sane) and same(got_bad, want_bad)
        assert bad.range_fallbacks == k + 1 and sane.range_fallbacks == 0
        if k == 1:
            assert bad.x3_range_events(reset=True) > 0         # a reset of one model's counters is invisible to the other
    assert sane.x3_range_events() == 0 and sane._twin is None
    assert same(f32_of_the_base_weights, want_bad)


def test_the_float32_twin_can_be_built_ahead_of_the_first_event():
    """ load_model(..., on_range_event='f32') builds its float32 twin inside the first affected call (weights upload, plan, tuning: seconds);
    prepare_fallback() does that ahead of time, and the host copy of the weights is released once the twin exists """
    model = models.load_model(scaled_weights(), backbone_name='resnet50', dtype='f16x3')
    assert model._twin is None and model._weights is not None
    model.prepare_fallback(B, H, Wd, 100, True)
    assert model._twin is not None and model._weights is None and (B, H, Wd, 100, True) in model._twin._plans
    out = model.predict_on_batch(inputs())
    assert model.range_fallbacks == 1 and int((out[2] > 0.05).sum()) > 0
