"""Per-stream view of ONE iteration from a rocprofv3 --kernel-trace CSV: python tools/stream_view.py <output dir> [k] [stream ...]
For every kernel of the chosen streams (default: all but the busiest one): start (ms since the iteration's first kernel), duration,
and the time since the previous kernel of the SAME stream ended -- on a dependent chain of small kernels that is queueing + launch
latency, i.e. what a high-priority side stream buys (or does not) while the main stream keeps the CUs full."""
import csv, sys, glob, collections
f = glob.glob(sys.argv[1] + '/*/*kernel_trace.csv')[0]
k = int(sys.argv[2]) if len(sys.argv) > 2 else 3
want = set(sys.argv[3:])
rows = []
for r in csv.DictReader(open(f)):
    rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], str(r.get('Stream_Id', r.get('Queue_Id', '?'))), int(r.get('Grid_Size_X', r.get('Grid_Size', 0)) or 0), int(r.get('Workgroup_Size_X', r.get('Workgroup_Size', 1)) or 1), str(r.get('Queue_Id', '?'))))
rows.sort()
ends = [r[0] for r in rows if 'adam_step_kernel' in r[2]]
t0, t1 = ends[-k - 2], ends[-k - 1]
it = [r for r in rows if t0 < r[0] <= t1]
busy = collections.Counter()
qmap = collections.defaultdict(set)
for s, e, n, q, g, w, hq in it:
    busy[q] += e - s
    qmap[q].add(hq)
print('hardware queue(s) of each stream:', {q: sorted(v) for q, v in qmap.items()})
print('iteration %.2f ms; kernel time by stream: %s' % ((t1 - t0) / 1e6, {q: round(v / 1e6, 2) for q, v in busy.items()}))
main = busy.most_common(1)[0][0]
def short(n):
    return n.replace('(anonymous namespace)::', '').replace('void ', '').replace('at::native::', '')[:60]
last = {}
for s, e, n, q, g, w, hq in it:
    if (want and q not in want) or (not want and q == main):
        continue
    wait = (s - last[q]) / 1e3 if q in last else 0.
    last[q] = e
    print(f'{q:>3} q{hq:>2} {(s - t0) / 1e6:8.3f} ms  dur {(e - s) / 1e3:8.1f} us  since prev {wait:8.1f} us  wg {g // max(w, 1):6d}  {short(n)}')
