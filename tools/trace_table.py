""" Per-launch table from a rocprofv3 --kernel-trace CSV of bench.py (last complete step). """
import csv
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'ground-plane-polling_amd'))
from keras_retinanet_3D.models import weights as W  # noqa: E402


def conv_names(backbone, fused_stages=(0, 1), block_stages=()):
    """ launch order of models/retinanet.py:_build (fused 2b+2c launches are bottleneck_tail_kernel, whole identity blocks bottleneck_block_x3_kernel) """
    names = []
    for stage, n in enumerate(W.BLOCKS[backbone]):
        for b in range(n):
            nm = W.block_name(backbone, stage, b)
            if b > 0 and stage in block_stages:
                names += ['res%s_2a+2b+2c' % nm]
                continue
            names += ['res%s_2a' % nm] + (['res%s_br1' % nm] if b == 0 else [])
            names += ['res%s_2b+2c' % nm] if stage in fused_stages else ['res%s_2b' % nm, 'res%s_2c' % nm]
    names += ['C5_reduced', 'P5', 'P6', 'P7', 'C4_reduced', 'P4', 'C3_reduced', 'P3']
    names += ['heads_0(fused)'] + ['cls_%d' % i for i in range(1, 4)] + ['cls_out']
    names += ['reg_%d' % i for i in range(1, 4)] + ['reg_ops'] + ['dim_%d' % i for i in range(1, 4)] + ['dim_out']
    return names


def short(k):
    m = re.search(r'bottleneck_block_x3_kernel<(\d+), (\d+), (\d+), (\d+), (\w+), (\d+)', k)
    if m:
        return 'whole block C=%s, %sx%s tiles, %s wavefronts%s' % (m.group(2), m.group(3), m.group(4), m.group(6), ', shortcut from the ring' if m.group(5) in ('true', '1') else '')
    m = re.search(r'bottleneck_tail_kernel<(\d+), (\d+), (\d+)>', k)
    if m:
        return 'fused tail %sx%s' % (m.group(2), m.group(3))
    if 'conv_igemm_dual_kernel' in k:
        return 'igemm 256x256 + 512x128 (dual grid)'
    m = re.search(r'conv_igemm_mix_kernel<(\d+), (\d+), (\d+)', k)
    if m:
        return 'igemm %sx256 + %sx256 (mixed heights)' % (m.group(2), m.group(3))
    m = re.search(r'conv1x1_ws_kernel<(\d+), (\d+), (\d+)', k)
    if m:
        return 'ws 1x1 %sx%s (weights resident)' % (m.group(2), m.group(3))
    m = re.search(r'conv_igemm_kernel<(\d+), (\d+), (\d+), (\d+), (\d+), (\d+), (\w+)(?:, (\w+))?>', k)
    if m:
        return 'igemm %sx%s w%sx%s%s%s' % (m.group(2), m.group(3), m.group(4), m.group(5), ' pipe' if m.group(7) in ('true', '1') else '',
                                         ' xin' if m.group(8) in ('true', '1') else '')
    for key in ('stem_pool_mfma', 'stem_mfma', 'stem_kernel', 'maxpool', 'relu', 'splitk_reduce', 'clear_counters', 'candidates', 'nms', 'emit_kernel', 'canonical_planes', 'poll'):
        if key in k:
            return key
    return k[:40]


def main(path, backbone='resnet50', fused='0,1', blocks=''):
    """ fused: stages whose 2b+2c run as one launch ('0,1' = the 16-bit default; '' for the float32-storage types); blocks: stages whose identity blocks
    run as one launch (x3 types, GPP_FUSE_BLOCK: '0,1' = res2 + res3) """
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    ours = [r for r in rows if 'at::native' not in r['Kernel_Name'] and 'rocclr' not in r['Kernel_Name']]
    ours = [r for r in ours if 'preprocess' not in r['Kernel_Name']]
    starts = [i for i, r in enumerate(ours) if 'stem' in r['Kernel_Name']]
    # complete steps = stem ... poll; bench.py's timed steps come first, the host-fed extra steps after them:
    # take the middle of the first 13 (warmup 3 + 10 timed with the profiling command of profiles/README.md)
    steps = [ours[a:b] for a, b in zip(starts, starts[1:] + [len(ours)]) if any('poll_kernel' in r['Kernel_Name'] for r in ours[a:b])]
    step = steps[min(8, len(steps) - 1)]
    names = conv_names(backbone, tuple(int(v) for v in fused.split(',') if v.strip()), tuple(int(v) for v in blocks.split(',') if v.strip()))
    ci = 0
    t0 = int(step[0]['Start_Timestamp'])
    total = 0.0
    groups = {}
    for r in step:
        k = r['Kernel_Name']
        d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
        total += d
        if 'conv_igemm' in k or 'bottleneck_tail' in k or 'bottleneck_block' in k or 'conv1x1_ws' in k:
            name = names[ci] if ci < len(names) else '?'
            ci += 1
        elif 'splitk_reduce' in k:
            name = '  (split-K reduce)'
        else:
            name = short(k)
        grp = 'stem+pool' if name in ('stem_pool_mfma', 'stem_mfma', 'stem_kernel', 'maxpool') else \
              'backbone' if name.startswith('res') else 'heads' if name[:3] in ('reg', 'dim', 'cls', 'hea') else \
              'decode+poll' if name in ('clear_counters', 'candidates', 'nms', 'emit_kernel', 'canonical_planes', 'poll') else \
              'fpn' if name[0] in 'CP' or name == 'relu' else 'other'
        if name.startswith('  ('):
            grp = last_grp
        last_grp = grp
        groups[grp] = groups.get(grp, 0.0) + d
        print('%-20s %-28s %9.1f us   +%.1f us' % (name, short(k), d, (int(r['Start_Timestamp']) - t0) / 1e3))
    wall = (int(step[-1]['End_Timestamp']) - t0) / 1e3
    print('sum of kernel durations %.1f us, wall of the step %.1f us' % (total, wall))
    print('by group: ' + ', '.join('%s %.0f us' % kv for kv in groups.items()))


if __name__ == '__main__':
    main(sys.argv[1], *(sys.argv[2:]))
