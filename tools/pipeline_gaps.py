""" From a rocprofv3 kernel trace of tools/bench_pipeline.py: GPU idle time between consecutive batches (end of one
batch's poll kernel -> start of the next batch's first kernel) and the busy span of a batch. """
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
polls = [r for r in rows if 'poll_kernel' in r['Kernel_Name']]
gaps, spans = [], []
for a, b in zip(polls, polls[1:]):
    end = int(a['End_Timestamp'])
    nxt = [r for r in rows if int(r['Start_Timestamp']) >= end and int(r['Start_Timestamp']) < int(b['Start_Timestamp'])]
    if not nxt:
        continue
    gaps.append((int(nxt[0]['Start_Timestamp']) - end) / 1e3)
    spans.append((int(b['End_Timestamp']) - int(nxt[0]['Start_Timestamp'])) / 1e3)
    first = nxt[0]['Kernel_Name'][:60]
print('batches', len(gaps), 'first kernel after a batch:', first)
gaps_s = sorted(gaps)
print('idle gap between batches (us): median %.1f  mean %.1f  p90 %.1f  max %.1f' % (gaps_s[len(gaps) // 2], sum(gaps) / len(gaps), gaps_s[int(len(gaps) * 0.9)], gaps_s[-1]))
spans_s = sorted(spans)
print('busy span of a batch (us): median %.1f  mean %.1f' % (spans_s[len(spans) // 2], sum(spans) / len(spans)))
