""" GPU numerics of the stem (7x7 s2 conv + folded BN + ReLU), the 'same' max-pool and ReLU
against plain PyTorch float32 references on the CPU. """
import pytest
import torch
import torch.nn.functional as F

from keras_retinanet_3D.backend import hip

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('dtype,tdt,eps', [(hip.GPP_BF16, torch.bfloat16, 2.0 ** -8), (hip.GPP_F16, torch.float16, 2.0 ** -10)])
@pytest.mark.parametrize('B,H,W', [(2, 37, 53), (1, 64, 131), (1, 402, 1333)])
def test_stem_matches_torch(B, H, W, dtype, tdt, eps):
    g = torch.Generator().manual_seed(H * W)
    x = torch.rand((B, H, W, 3), generator=g) * 255.0 - 120.0
    k = torch.randn((7, 7, 3, 64), generator=g) * 0.05
    bias = torch.randn((64,), generator=g)
    ref = torch.relu(F.conv2d(F.pad(x.permute(0, 3, 1, 2), (3, 3, 3, 3)), k.permute(3, 2, 0, 1), bias, stride=2)).permute(0, 2, 3, 1)
    Ho, Wo = ref.shape[1:3]
    dev = torch.device('cuda')
    out = torch.full((B, Ho, Wo, 64), float('nan'), dtype=tdt, device=dev)
    xd, kd, bd = x.to(dev).contiguous(), k.reshape(147, 64).to(dev).contiguous(), bias.to(dev)
    hip.check(hip.lib().gpp_stem_conv7x7_bn_relu(hip.ptr(xd), hip.ptr(kd), hip.ptr(bd), hip.ptr(out), dtype, B, H, W, hip.stream_ptr()))
    got = out.float().cpu()
    err = (got - ref).abs()
    assert bool((err <= eps * ref.abs() + 2e-3).all()), err.max().item()

    # MFMA stem (the one the model uses): input and weights rounded to f16, float32 accumulation.
    # Rigorous bound: every product carries <= 2 * 2^-11 relative rounding -> 2^-10 * sum |x||w|.
    bound = F.conv2d(F.pad(x.abs().permute(0, 3, 1, 2), (3, 3, 3, 3)), k.abs().permute(3, 2, 0, 1), None, stride=2).permute(0, 2, 3, 1)
    out2 = torch.full((B, Ho, Wo, 64), float('nan'), dtype=tdt, device=dev)
    packed = hip.pack_stem_weights(k.reshape(147, 64).numpy(), dev)
    hip.check(hip.lib().gpp_stem_conv7x7_bn_relu_mfma(hip.ptr(xd), hip.ptr(packed), hip.ptr(bd), hip.ptr(out2), dtype, B, H, W, hip.stream_ptr()))
    got2 = out2.float().cpu()
    assert torch.isfinite(got2).all()
    err2 = (got2 - ref).abs()
    assert bool((err2 <= eps * ref.abs() + 2.0 ** -10 * bound + 2e-3).all()), err2.max().item()
    assert err2.mean().item() < 0.1 * (2.0 ** -10 * bound).mean().item() + eps * ref.abs().mean().item()


@pytest.mark.parametrize('B,H,W', [(2, 37, 53), (1, 64, 131), (3, 17, 300), (2, 402, 1333)])
def test_x3_stem_matches_float64(B, H, W):
    """ the matrix-pipe stem of the float32-storage x3 types (gpp_stem_conv7x7_bn_relu_x3): pixels and weights split into two IEEE
    halves each, three products per float32 product, per-channel power-of-two weight scale.  Against float64 on the float32 operands:
    |err| <= 1e-5 |ref| + 2e-6 rms(ref) sqrt(147) (the float32 conv bar of tests/test_conv_f32_gpu.py) and rms(err) <= 5e-7 rms(ref);
    channels whose weights are four decades apart keep that relative accuracy; ragged map edges and tiles (8-row workgroup tiles) """
    g = torch.Generator().manual_seed(H * W + B)
    x = torch.rand((B, H, W, 3), generator=g) * 255.0 - 120.0
    k = torch.randn((7, 7, 3, 64), generator=g) * 0.05 * torch.pow(10.0, torch.linspace(-2.0, 2.0, 64))[None, None, None, :]
    bias = torch.randn((64,), generator=g) * 0.1
    ref = torch.relu(F.conv2d(F.pad(x.double().permute(0, 3, 1, 2), (3, 3, 3, 3)), k.double().permute(3, 2, 0, 1), bias.double(), stride=2)).permute(0, 2, 3, 1)
    pre = F.conv2d(F.pad(x.double().permute(0, 3, 1, 2), (3, 3, 3, 3)), k.double().permute(3, 2, 0, 1), None, stride=2).permute(0, 2, 3, 1)
    Ho, Wo = ref.shape[1:3]
    dev = torch.device('cuda')
    out = torch.full((B, Ho, Wo, 64), float('nan'), dtype=torch.float32, device=dev)
    xd, bd = x.to(dev).contiguous(), bias.to(dev)
    packed = hip.pack_stem_weights_x3(k.reshape(147, 64).numpy(), dev)
    hip.check(hip.lib().gpp_stem_conv7x7_bn_relu_x3(hip.ptr(xd), hip.ptr(packed), hip.ptr(bd), hip.ptr(out), B, H, W, hip.stream_ptr()))
    got = out.double().cpu()
    assert torch.isfinite(got).all()
    rms_c = pre.reshape(-1, 64).pow(2).mean(dim=0).sqrt()                    # per channel: the channels differ by four decades
    err = (got - ref).abs().reshape(-1, 64)
    tol = 1e-5 * ref.abs().reshape(-1, 64) + 2e-6 * 147 ** 0.5 * rms_c[None, :]
    assert bool((err <= tol).all()), (err / rms_c[None, :]).max().item()
    assert bool((err.pow(2).mean(dim=0).sqrt() <= 5e-7 * rms_c + 1e-8).all()), (err.pow(2).mean(dim=0).sqrt() / rms_c).max().item()
    assert hip.lib().gpp_stem_conv7x7_bn_relu_x3(None, hip.ptr(packed), hip.ptr(bd), hip.ptr(out), B, H, W, hip.stream_ptr()) == -1


@pytest.mark.parametrize('B,H,W,C', [(2, 19, 27, 64), (1, 20, 28, 64), (1, 201, 667, 64)])
def test_maxpool_matches_torch(B, H, W, C):
    g = torch.Generator().manual_seed(H)
    x = torch.randn((B, H, W, C), generator=g).to(torch.bfloat16)
    Ho, Wo = (H + 1) // 2, (W + 1) // 2
    pt = max((Ho - 1) * 2 + 3 - H, 0)
    pl = max((Wo - 1) * 2 + 3 - W, 0)
    xp = F.pad(x.float().permute(0, 3, 1, 2), (pl // 2, pl - pl // 2, pt // 2, pt - pt // 2), value=float('-inf'))
    ref = F.max_pool2d(xp, 3, 2).permute(0, 2, 3, 1)
    dev = torch.device('cuda')
    out = torch.empty((B, Ho, Wo, C), dtype=torch.bfloat16, device=dev)
    xd = x.to(dev).contiguous()
    hip.check(hip.lib().gpp_maxpool3x3s2_same(hip.ptr(xd), hip.ptr(out), hip.GPP_BF16, B, H, W, C, hip.stream_ptr()))
    assert torch.equal(out.float().cpu(), ref)


@pytest.mark.parametrize('dtype,tdt', [(hip.GPP_BF16, torch.bfloat16), (hip.GPP_F16, torch.float16)])
@pytest.mark.parametrize('B,H,W', [(2, 37, 53), (1, 40, 56), (3, 38, 55), (1, 7, 9), (2, 96, 160), (1, 127, 211), (1, 270, 500), (2, 402, 1333)])
def test_fused_stem_pool_equals_the_two_launches(B, H, W, dtype, tdt):
    """ conv1 + bn + relu + pool1 in one launch == the MFMA stem followed by the max-pool, bit for bit: odd and even conv map
    sizes (pool padding 1 / 0 on either axis, i.e. with and without the extra first row of a strip), maps narrower than one
    strip and shorter than one row block, several strips / row blocks / images per persistent workgroup. """
    g = torch.Generator().manual_seed(H * W + B)
    x = torch.rand((B, H, W, 3), generator=g) * 255.0 - 120.0
    k = torch.randn((7, 7, 3, 64), generator=g) * 0.05
    bias = torch.randn((64,), generator=g)
    dev = torch.device('cuda')
    Ho, Wo = (H + 6 - 7) // 2 + 1, (W + 6 - 7) // 2 + 1
    Hp, Wp = (Ho + 1) // 2, (Wo + 1) // 2
    xd, bd = x.to(dev).contiguous(), bias.to(dev)
    packed = hip.pack_stem_weights(k.reshape(147, 64).numpy(), dev)
    conv = torch.full((B, Ho, Wo, 64), float('nan'), dtype=tdt, device=dev)
    want = torch.full((B, Hp, Wp, 64), float('nan'), dtype=tdt, device=dev)
    got = torch.full((B, Hp, Wp, 64), float('nan'), dtype=tdt, device=dev)
    hip.check(hip.lib().gpp_stem_conv7x7_bn_relu_mfma(hip.ptr(xd), hip.ptr(packed), hip.ptr(bd), hip.ptr(conv), dtype, B, H, W, hip.stream_ptr()))
    hip.check(hip.lib().gpp_maxpool3x3s2_same(hip.ptr(conv), hip.ptr(want), dtype, B, Ho, Wo, 64, hip.stream_ptr()))
    hip.check(hip.lib().gpp_stem_pool_fused_mfma(hip.ptr(xd), hip.ptr(packed), hip.ptr(bd), hip.ptr(got), dtype, B, H, W, hip.stream_ptr()))
    torch.cuda.synchronize()
    a, b = got.view(torch.int16).cpu(), want.view(torch.int16).cpu()
    assert not torch.isnan(want.float()).any()
    assert torch.equal(a, b), 'differs at {} of {} elements'.format(int((a != b).sum()), a.numel())


@pytest.mark.parametrize('big', [False, True])
@pytest.mark.parametrize('B,H,W', [(2, 37, 53), (1, 40, 56), (3, 38, 55), (1, 7, 9), (2, 96, 160), (1, 127, 211), (1, 270, 500), (2, 402, 1333), (1, 250, 249), (1, 252, 251)])
def test_fused_x3_stem_pool_equals_the_two_launches(B, H, W, big):
    """ gpp_stem_pool_fused_x3 (conv1 + bn + relu + pool1 of the x3 types in one launch, the horizontal half of the pool taken in the accumulator
    lanes) == gpp_stem_conv7x7_bn_relu_x3_rc followed by gpp_maxpool3x3s2_same(GPP_F32), bit for bit, on the shapes of the 16-bit test plus
    maps whose last strip is full (Wp = 62 / 63); and it counts the range events of the conv map exactly as the unfused stem does
    (`big`: weights that drive part of the conv map beyond 65504 -- in the overlap columns of two strips and in carried rows too) """
    g = torch.Generator().manual_seed(H * W + B)
    x = torch.rand((B, H, W, 3), generator=g) * 255.0 - 120.0
    k = torch.randn((7, 7, 3, 64), generator=g) * (40.0 if big else 0.05)
    bias = torch.randn((64,), generator=g)
    dev = torch.device('cuda')
    Ho, Wo = (H + 6 - 7) // 2 + 1, (W + 6 - 7) // 2 + 1
    Hp, Wp = (Ho + 1) // 2, (Wo + 1) // 2
    xd, bd = x.to(dev).contiguous(), bias.to(dev)
    packed = hip.pack_stem_weights_x3(k.reshape(147, 64).numpy(), dev)
    conv = torch.full((B, Ho, Wo, 64), float('nan'), dtype=torch.float32, device=dev)
    want = torch.full((B, Hp, Wp, 64), float('nan'), dtype=torch.float32, device=dev)
    got = torch.full((B, Hp, Wp, 64), float('nan'), dtype=torch.float32, device=dev)
    slots = torch.zeros((2,), dtype=torch.int64, device=dev)
    lib = hip.lib()
    hip.check(lib.gpp_stem_conv7x7_bn_relu_x3_rc(hip.ptr(xd), hip.ptr(packed), hip.ptr(bd), hip.ptr(conv), B, H, W, slots.data_ptr(), hip.stream_ptr()))
    hip.check(lib.gpp_maxpool3x3s2_same(hip.ptr(conv), hip.ptr(want), hip.GPP_F32, B, Ho, Wo, 64, hip.stream_ptr()))
    hip.check(lib.gpp_stem_pool_fused_x3(hip.ptr(xd), hip.ptr(packed), hip.ptr(bd), hip.ptr(got), B, H, W, slots.data_ptr() + 8, hip.stream_ptr()))
    torch.cuda.synchronize()
    a, b = got.view(torch.int32).cpu(), want.view(torch.int32).cpu()
    assert not torch.isnan(want).any()
    assert torch.equal(a, b), 'differs at {} of {} elements'.format(int((a != b).sum()), a.numel())
    counts = slots.cpu().tolist()
    assert counts[0] == counts[1] and (counts[0] > 0) == big, counts
    if big:                # one event per (pixel, group of 8 channels = what one lane stores) of the conv map with a value beyond the half range
        assert counts[0] == int((conv.view(B, Ho, Wo, 8, 8) > 65504.0).any(dim=-1).sum())


def test_fused_x3_stem_pool_four_row_form_in_a_child_process():
    """ GPP_STEM_POOL_X3_ROWS=4 (4 wavefronts, 4 conv rows per step; read once per process): the measured alternative of the shipped 6-row form gives the
    same bytes -- tools/bench_stem.py in a child process on a map of several strips and row blocks """
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, GPP_STEM_POOL_X3_ROWS='4')
    out = subprocess.run([sys.executable, os.path.join(root, 'tools', 'bench_stem.py'), '2', '150', '333'], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                         universal_newlines=True, timeout=300).stdout
    assert 'bytes equal: True' in out and 'GPP_STEM_POOL_X3_ROWS=4' in out, out[-800:]


def test_fused_x3_stem_pool_arguments():
    dev = torch.device('cuda')
    x = torch.zeros((1, 16, 16, 3), device=dev)
    w = torch.zeros((2 * 64 * 232 + 128,), dtype=torch.float16, device=dev)
    b = torch.zeros((64,), device=dev)
    out = torch.zeros((1, 4, 4, 64), device=dev)
    lib = hip.lib()
    assert lib.gpp_stem_pool_fused_x3(hip.ptr(x), hip.ptr(w), hip.ptr(b), hip.ptr(out), 1, 16, 16, None, hip.stream_ptr()) == 0
    assert lib.gpp_stem_pool_fused_x3(None, hip.ptr(w), hip.ptr(b), hip.ptr(out), 1, 16, 16, None, hip.stream_ptr()) == -1
    assert lib.gpp_stem_pool_fused_x3(hip.ptr(x), hip.ptr(w), hip.ptr(b), hip.ptr(out), 0, 16, 16, None, hip.stream_ptr()) == -1
    assert lib.gpp_stem_pool_fused_x3(hip.ptr(x), hip.ptr(w), hip.ptr(b), out.data_ptr() + 4, 1, 16, 16, None, hip.stream_ptr()) == -3
    assert lib.gpp_stem_pool_fused_x3(hip.ptr(x), hip.ptr(w), hip.ptr(b), hip.ptr(out), 1, 16, 16, out.data_ptr() + 4, hip.stream_ptr()) == -3
    torch.cuda.synchronize()


def test_fused_stem_pool_rejects_float32():
    dev = torch.device('cuda')
    x = torch.zeros((1, 16, 16, 3), device=dev)
    w = torch.zeros((64 * 232,), dtype=torch.float16, device=dev)
    b = torch.zeros((64,), device=dev)
    out = torch.zeros((1, 4, 4, 64), device=dev)
    assert hip.lib().gpp_stem_pool_fused_mfma(hip.ptr(x), hip.ptr(w), hip.ptr(b), hip.ptr(out), hip.GPP_F32, 1, 16, 16, hip.stream_ptr()) == -4
    assert hip.lib().gpp_stem_pool_fused_mfma(None, hip.ptr(w), hip.ptr(b), hip.ptr(out), hip.GPP_BF16, 1, 16, 16, hip.stream_ptr()) == -1


def test_relu():
    dev = torch.device('cuda')
    x = torch.randn((3, 7, 21, 512)).to(torch.bfloat16).to(dev)
    out = torch.empty_like(x)
    hip.check(hip.lib().gpp_relu(hip.ptr(x), hip.ptr(out), hip.GPP_BF16, x.numel(), hip.stream_ptr()))
    assert torch.equal(out, torch.relu(x))


def test_gpu_preprocessing_is_bit_identical_to_the_host_path():
    """ uint8 frames -> (mean subtraction, bilinear resize) on the device == utils.image on the host """
    import numpy as np
    from keras_retinanet_3D import models
    from keras_retinanet_3D.utils import image, synthetic
    model = models.load_model('synthetic:3', backbone_name='resnet50', dtype='bf16')
    frames = np.stack([synthetic.synthetic_image(seed=s) for s in (1, 2)])
    planes = synthetic.load_plane_database('10').astype(np.float32)
    scale = image.compute_resize_scale(frames.shape[1:])
    _, P_inv = synthetic.synthetic_calibration(scale)
    P_inv = np.tile(P_inv[None].astype(np.float32), (2, 1, 1))
    plan, got_scale = model.stage_frames(frames, P_inv, planes)
    want = np.stack([image.resize_image(image.preprocess_image(f))[0] for f in frames])
    assert got_scale == scale and tuple(plan.images.shape) == want.shape == (2, 402, 1333, 3)
    assert np.array_equal(plan.images.cpu().numpy(), want)
    a, _ = model.predict_on_frames(frames, P_inv, planes)
    b = model.predict_on_batch([want, P_inv, planes])
    for x, y in zip(a, b):
        assert np.array_equal(x, y) or np.allclose(x, y, equal_nan=True)
