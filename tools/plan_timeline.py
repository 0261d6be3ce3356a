""" Timeline of ONE step of the default (multi-stream) plan from a rocprofv3 --kernel-trace CSV of bench.py: every launch with its queue,
start offset, duration and grid, what overlaps what, the wall of the step and the time during which nothing runs.
    python tools/plan_timeline.py <kernel_trace.csv> [step index, default 8] """
import csv
import re
import sys


def short(k):
    m = re.search(r'conv_igemm_kernel<(\d+), (\d+), (\d+), (\d+), (\d+), (\d+), (\w+)(?:, (\w+))?>', k)
    if m:
        return 'igemm {}x{}{}'.format(m.group(2), m.group(3), ' pipe' if m.group(7) in ('true', '1') else '')
    m = re.search(r'conv_igemm_mix_kernel<(\d+), (\d+), (\d+)', k)
    if m:
        return 'igemm mix {}+{}'.format(m.group(2), m.group(3))
    m = re.search(r'conv1x1_ws_kernel<(\d+), (\d+), (\d+)', k)
    if m:
        return 'ws 1x1 {}x{}'.format(m.group(2), m.group(3))
    for key in ('conv_igemm_dual', 'bottleneck_tail_x3', 'bottleneck_tail', 'stem_pool_mfma', 'stem_mfma_x3', 'stem_mfma', 'maxpool', 'relu', 'splitk_reduce',
                'clear_counters', 'candidates', 'nms', 'emit_kernel', 'canonical_planes', 'poll'):
        if key in k:
            return key
    return k[:40]


rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
ours = [r for r in rows if 'at::native' not in r['Kernel_Name'] and 'rocclr' not in r['Kernel_Name'] and 'preprocess' not in r['Kernel_Name']]
starts = [i for i, r in enumerate(ours) if 'stem' in r['Kernel_Name']]
steps = [ours[a:b] for a, b in zip(starts, starts[1:] + [len(ours)]) if any('poll_kernel' in r['Kernel_Name'] for r in ours[a:b])]
step = steps[min(int(sys.argv[2]) if len(sys.argv) > 2 else 8, len(steps) - 1)]
t0 = int(step[0]['Start_Timestamp'])
queues = {}
events = []
for r in step:
    q = queues.setdefault(r.get('Queue_Id', '?'), len(queues))
    s, e = (int(r['Start_Timestamp']) - t0) / 1e3, (int(r['End_Timestamp']) - t0) / 1e3
    wg = int(r.get('Workgroup_Size_X', r.get('Workgroup_Size', 1)) or 1)
    grid = int(r.get('Grid_Size_X', r.get('Grid_Size', 0)) or 0) // max(wg, 1)
    events.append((s, e, q, short(r['Kernel_Name']), grid))
wall = max(e for _, e, _, _, _ in events)
busy, cur_s, cur_e = 0.0, None, None
for s, e, _, _, _ in sorted(events):
    if cur_e is None or s > cur_e:
        if cur_e is not None:
            busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
for s, e, q, name, grid in events:
    others = [n for s2, e2, q2, n, _ in events if q2 != q and s2 < e and e2 > s]
    print('q{} +{:8.1f} us {:7.1f} us {:6d} WGs  {:24s} {}'.format(q, s, e - s, grid, name, ('|| ' + ', '.join(others[:3])) if others else ''))
print('wall of the step {:.1f} us, some kernel running {:.1f} us, nothing running {:.1f} us, sum of durations {:.1f} us'.format(
    wall, busy, wall - busy, sum(e - s for s, e, _, _, _ in events)))
