"""Condensed per-queue timeline of one replayed frame from a rocprofv3 kernel trace:
python tools/frame_timeline.py <kernel_trace.csv> [frame_index] [--full]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'render_rays_kernel' in r['Kernel_Name']]
k = int(sys.argv[2]) if len(sys.argv) > 2 and not sys.argv[2].startswith('-') else len(idx) // 2
fr = rows[idx[k]:idx[k + 1]]
t0 = int(fr[0]['Start_Timestamp'])
wall = (int(rows[idx[k + 1]]['Start_Timestamp']) - t0) / 1e3
iv = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp'])) for r in fr)
busy, (cs, ce) = 0, iv[0]
for s, e in iv[1:]:
    if s > ce:
        busy += ce - cs
        cs, ce = s, e
    else:
        ce = max(ce, e)
busy += ce - cs
print(f'frame {k}: wall {wall:.1f} us, {len(fr)} kernels, union busy {busy / 1e3:.1f} us, sum of durations {sum(e - s for s, e in iv) / 1e3:.1f} us')
queues = {}
for r in fr:
    q = r['Queue_Id']
    s, e = (int(r['Start_Timestamp']) - t0) / 1e3, (int(r['End_Timestamp']) - t0) / 1e3
    name = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0]
    queues.setdefault(q, []).append((s, e, name))
for q, ks in sorted(queues.items()):
    # merge into spans separated by gaps > 30 us
    spans, cur = [], [ks[0][0], ks[0][1], 1, ks[0][1] - ks[0][0]]
    for s, e, _ in ks[1:]:
        if s - cur[1] > 30:
            spans.append(cur)
            cur = [s, e, 1, e - s]
        else:
            cur[1] = max(cur[1], e); cur[2] += 1; cur[3] += e - s
    spans.append(cur)
    print(f'queue {q}: {len(ks)} kernels')
    for s, e, n, b in spans:
        print(f'    {s:8.1f} .. {e:8.1f}  ({e - s:7.1f} us, {n:3d} kernels, busy {b:7.1f})')
if '--full' in sys.argv:
    for r in fr:
        s, e = (int(r['Start_Timestamp']) - t0) / 1e3, (int(r['End_Timestamp']) - t0) / 1e3
        name = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '')[:70]
        print(f"{s:9.1f} {e:9.1f} {e - s:8.1f} q{r['Queue_Id']} {name}")
