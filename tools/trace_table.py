""" Per-layer table from a rocprofv3 --kernel-trace CSV of bench.py (last complete step). """
import csv
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'ground-plane-polling_amd'))
from keras_retinanet_3D.models import weights as W  # noqa: E402


def op_names(backbone):
    names = ['conv1(stem)', 'pool1']
    for stage, n in enumerate(W.BLOCKS[backbone]):
        for b in range(n):
            nm = W.block_name(backbone, stage, b)
            names += ['res%s_2a' % nm, 'res%s_2b' % nm] + (['res%s_br1' % nm] if b == 0 else []) + ['res%s_2c' % nm]
    names += ['C5_reduced', 'P5', 'C4_reduced', 'P4', 'C3_reduced', 'P3', 'P6', 'C6_relu', 'P7']
    names += ['reg_%d' % i for i in range(4)] + ['reg_ops'] + ['dim_%d' % i for i in range(4)] + ['dim_out']
    names += ['cls_%d' % i for i in range(4)] + ['cls_out', 'memset', 'candidates', 'nms', 'copy_counts', 'canon_planes', 'poll']
    return names


def main(path, backbone='resnet50'):
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    ours = [r for r in rows if 'at::native' not in r['Kernel_Name']]
    # steps start at every stem kernel
    starts = [i for i, r in enumerate(ours) if 'stem_kernel' in r['Kernel_Name']]
    i0 = starts[-1]
    step = ours[i0:]
    names = [n for n in op_names(backbone) if n not in ('memset', 'copy_counts')]
    step = [r for r in step if 'rocclr' not in r['Kernel_Name']][:len(names)]
    tot = 0.0
    t0 = int(step[0]['Start_Timestamp'])
    for n, r in zip(names, step):
        d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
        tot += d
        k = r['Kernel_Name']
        short = 'igemm128x128' if '128, 128' in k else 'igemm128x64' if '128, 64' in k else k.split('(')[0][-30:]
        print('%-16s %-30s %9.1f us   grid %s  start +%.1f us' % (n, short, d, r.get('Grid_Size', ''), (int(r['Start_Timestamp']) - t0) / 1e3))
    wall = (int(step[-1]['End_Timestamp']) - t0) / 1e3
    print('sum of kernel durations %.1f us, wall of the step %.1f us' % (tot, wall))


if __name__ == '__main__':
    main(sys.argv[1], *(sys.argv[2:]))
