"""
ORACLE (test infrastructure, not product code): PyTorch-CPU restatement of the RetinaNet-3D
forward graph (ResNet-50/101/152 + FPN + three heads) in float32.

PARITY UNPINNED for this stage: the reference defines the FPN and the heads
(/root/reference/keras_retinanet_3D/models/retinanet.py:24-205,257-281) but the backbone is the
third-party, unpinned `keras_resnet` package (setup.py:7; constructor call models/resnet.py:88-93)
on Keras/TF1 (README.md:12), none of which exist here, and the reference ships no tests, weights or
golden outputs for it.  The backbone below restates keras_resnet's published architecture
(ResNet2D / bottleneck_2d: ZeroPadding2D(3), conv1 7x7/2 valid no-bias, bn_conv1 eps 1e-5, ReLU,
MaxPooling2D(3, stride 2, 'same'); bottleneck = 1x1 (stride on this conv) / BN / ReLU /
ZeroPadding2D(1) + 3x3 valid / BN / ReLU / 1x1 x4 / BN, projection shortcut on block 0, Add, ReLU;
freeze_bn => moving statistics), and TF semantics for 'same' padding and nearest resize.

Three modes
  * float32 throughout, BatchNormalization applied literally (the reference semantics);
  * `precision='f64'`: the same literal graph in float64 -- the EXACT value of what the reference's float32 graph approximates
    (head tensors rounded to float32 once, at the end): the yardstick that tells whether an arithmetic mode is as close to the
    true result as float32 itself is (oracle/gen_fullsize_goldens.py);
  * `storage` = 'bf16' | 'f16': BN folded into the convolution, weights and every stored
    activation rounded to the 16-bit storage type of the HIP path (the stem rounds image and
    weights to f16), float32 accumulation -- the arithmetic the GPU performs, up to summation order.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""

import numpy as np
import torch
import torch.nn.functional as F

BN_EPS = 1e-5
BLOCKS = {'resnet50': (3, 4, 6, 3), 'resnet101': (3, 4, 23, 3), 'resnet152': (3, 8, 36, 3)}
NUMERICAL = {'resnet50': (0, 0, 0, 0), 'resnet101': (0, 1, 1, 0), 'resnet152': (0, 1, 1, 0)}
NUM_ANCHORS = 12


def _quantizer(storage):
    if storage is None:
        return lambda t: t
    dt = {'bf16': torch.bfloat16, 'f16': torch.float16}[storage]
    return lambda t: t.to(dt).to(torch.float32)


def _same_pad(size, k, s):
    out = -(-size // s)
    total = max((out - 1) * s + k - size, 0)
    return total // 2, total - total // 2


def _conv(x, kernel_hwio, stride=1, padding='valid', pad=None):
    """ x NCHW float32; Keras HWIO kernel; TF 'same' (extra padding after) or explicit/valid """
    k = torch.as_tensor(kernel_hwio).permute(3, 2, 0, 1).contiguous()
    if padding == 'same':
        ph = _same_pad(x.shape[2], k.shape[2], stride)
        pw = _same_pad(x.shape[3], k.shape[3], stride)
        x = F.pad(x, (pw[0], pw[1], ph[0], ph[1]))
    elif pad:
        x = F.pad(x, (pad, pad, pad, pad))
    return F.conv2d(x, k, None, stride=stride)


def _nearest_like(src, target):
    """ layers.UpsampleLike: tf.image.resize_images(NEAREST, align_corners=False) """
    ih, iw = src.shape[2:]
    oh, ow = target.shape[2:]
    ys = torch.clamp(torch.floor(torch.arange(oh, dtype=torch.float32) * (np.float32(ih) / np.float32(oh))).long(), max=ih - 1)
    xs = torch.clamp(torch.floor(torch.arange(ow, dtype=torch.float32) * (np.float32(iw) / np.float32(ow))).long(), max=iw - 1)
    # (index arithmetic stays float32 in every mode: it is TF's, not the graph's floatx)
    return src[:, :, ys][:, :, :, xs]


class Net(object):
    def __init__(self, weights, backbone='resnet50', storage=None, precision='f32'):
        assert precision in ('f32', 'f64') and not (storage and precision == 'f64')
        self.dt = torch.float64 if precision == 'f64' else torch.float32
        self.w = {k: np.asarray(v, np.float64) for k, v in weights.items()} if precision == 'f64' else weights
        self.backbone = backbone
        self.storage = storage
        self.q = _quantizer(storage)
        self.trace = None            # when a dict: every layer output, {(layer name, pyramid level): NHWC array}
        self._level = 0

    def _rec(self, name, y):
        if self.trace is not None:
            self.trace[(name, self._level)] = y.permute(0, 2, 3, 1).contiguous().numpy()
        return y

    # conv + frozen BN (+ ReLU); literal BN in float32 mode, folded + rounded weights in storage mode
    def conv_bn(self, x, conv, bn, stride=1, pad=None, relu=True, add=None, quant_weights=True, f16_operands=False):
        w = self.w
        k = torch.as_tensor(w[conv + '/kernel'])
        gamma, beta = torch.as_tensor(w[bn + '/gamma']), torch.as_tensor(w[bn + '/beta'])
        mean, var = torch.as_tensor(w[bn + '/moving_mean']), torch.as_tensor(w[bn + '/moving_variance'])
        if self.storage is None:
            y = _conv(x, k, stride=stride, pad=pad)
            y = (y - mean[None, :, None, None]) / torch.sqrt(var[None, :, None, None] + BN_EPS) * gamma[None, :, None, None] \
                + beta[None, :, None, None]
        else:
            s = gamma.double() / torch.sqrt(var.double() + BN_EPS)
            kf = (k.double() * s[None, None, None, :]).float()
            bf = (beta.double() - mean.double() * s).float()
            if f16_operands:      # the MFMA stem rounds image and weights to f16 (csrc/stem.hip), float32 accumulation
                x, kf = x.half().float(), kf.half().float()
            y = _conv(x, self.q(kf) if quant_weights else kf, stride=stride, pad=pad) + bf[None, :, None, None]
        if add is not None:
            y = y + add
        if relu:
            y = torch.relu(y)
        return self._rec(conv, self.q(y))

    def conv_bias(self, x, name, stride=1, relu=False, add=None, store=True):
        k = torch.as_tensor(self.w[name + '/kernel'])
        b = torch.as_tensor(self.w[name + '/bias'])
        y = _conv(x, self.q(k), stride=stride, padding='same') + b[None, :, None, None]
        if add is not None:
            y = y + add
        if relu:
            y = torch.relu(y)
        return self._rec(name, self.q(y) if store else y)

    def block_name(self, stage, block):
        if block > 0 and NUMERICAL[self.backbone][stage]:
            return '{}b{}'.format(stage + 2, block)
        return '{}{}'.format(stage + 2, chr(ord('a') + block))

    def resnet(self, x):
        x = self.conv_bn(x, 'conv1', 'bn_conv1', stride=2, pad=3, quant_weights=False, f16_operands=True)
        ph = _same_pad(x.shape[2], 3, 2)
        pw = _same_pad(x.shape[3], 3, 2)
        x = self._rec('pool1', F.max_pool2d(F.pad(x, (pw[0], pw[1], ph[0], ph[1]), value=float('-inf')), 3, 2))
        outs = []
        for stage, n_blocks in enumerate(BLOCKS[self.backbone]):
            for block in range(n_blocks):
                nm = self.block_name(stage, block)
                stride = 2 if (block == 0 and stage > 0) else 1
                y = self.conv_bn(x, 'res{}_branch2a'.format(nm), 'bn{}_branch2a'.format(nm), stride=stride)
                y = self.conv_bn(y, 'res{}_branch2b'.format(nm), 'bn{}_branch2b'.format(nm), pad=1)
                if block == 0:
                    sc = self.conv_bn(x, 'res{}_branch1'.format(nm), 'bn{}_branch1'.format(nm), stride=stride, relu=False)
                else:
                    sc = x
                x = self.conv_bn(y, 'res{}_branch2c'.format(nm), 'bn{}_branch2c'.format(nm), relu=True, add=sc)
            outs.append(x)
        return outs          # C2, C3, C4, C5

    def fpn(self, C3, C4, C5):
        """ models/retinanet.py:170-205 """
        P5 = self.conv_bias(C5, 'C5_reduced')
        P5_up = _nearest_like(P5, C4)
        P5 = self.conv_bias(P5, 'P5')
        P4 = self.conv_bias(C4, 'C4_reduced', add=P5_up)
        P4_up = _nearest_like(P4, C3)
        P4 = self.conv_bias(P4, 'P4')
        P3 = self.conv_bias(C3, 'C3_reduced', add=P4_up)
        P3 = self.conv_bias(P3, 'P3')
        P6 = self.conv_bias(C5, 'P6', stride=2)
        P7 = self.conv_bias(self._rec('C6_relu', self.q(torch.relu(P6))), 'P7', stride=2)
        return [P3, P4, P5, P6, P7]

    @staticmethod
    def _rows(y, k):
        """ keras Reshape((-1, k)) of an NHWC tensor """
        return y.permute(0, 2, 3, 1).reshape(y.shape[0], -1, k)

    def heads(self, features):
        """ models/retinanet.py:24-167 applied per level and concatenated along axis 1 (:257-281) """
        reg, dim, cls = [], [], []
        for level, f in enumerate(features):
            self._level = level
            y = f
            for i in range(4):
                y = self.conv_bias(y, 'pyramid_regression_{}'.format(i), relu=True)
            parts = [self._rows(self.conv_bias(y, 'pyramid_regression_op1', store=False), 4)]
            parts += [self._rows(self.conv_bias(y, 'pyramid_regression_op{}'.format(k), store=False), 2) for k in (2, 3, 4, 5)]
            reg.append(torch.cat(parts, dim=2))
            y = f
            for i in range(4):
                y = self.conv_bias(y, 'pyramid_regression_dim_{}'.format(i), relu=True)
            dim.append(self._rows(self.conv_bias(y, 'pyramid_regression_dim', store=False), 3))
            y = f
            for i in range(4):
                y = self.conv_bias(y, 'pyramid_classification_{}'.format(i), relu=True)
            cls.append(self._rows(self.conv_bias(y, 'pyramid_classification', store=False), 8))
        self._level = 0
        return torch.cat(reg, dim=1), torch.cat(dim, dim=1), torch.cat(cls, dim=1)

    def forward(self, images_nhwc, keep_features=False, trace=False):
        """ images (B, H, W, 3) float32 BGR mean-subtracted -> dict of NumPy arrays:
        regression (B, A, 12), regression_dim (B, A, 3), classification_logits (B, A, 8) """
        self.trace = {} if trace else None
        with torch.no_grad():
            x = torch.as_tensor(np.ascontiguousarray(images_nhwc, dtype=np.float32)).permute(0, 3, 1, 2).to(self.dt)
            C2, C3, C4, C5 = self.resnet(x)
            feats = self.fpn(C3, C4, C5)
            reg, dim, cls = self.heads(feats)
        out = {'regression': reg.float().numpy(), 'regression_dim': dim.float().numpy(), 'classification_logits': cls.float().numpy()}
        if trace:
            out['trace'] = self.trace
        if keep_features:
            for name, t in zip(('C2', 'C3', 'C4', 'C5', 'P3', 'P4', 'P5', 'P6', 'P7'), [C2, C3, C4, C5] + feats):
                out[name] = t.permute(0, 2, 3, 1).contiguous().numpy()
        return out


def forward(weights, images_nhwc, backbone='resnet50', storage=None, keep_features=False, trace=False, precision='f32'):
    return Net(weights, backbone, storage, precision).forward(images_nhwc, keep_features=keep_features, trace=trace)
