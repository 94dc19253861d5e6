"""Timeline of ONE training iteration from a rocprofv3 --kernel-trace CSV: python tools/iteration_timeline.py <output dir> [k]
Iterations are delimited by Adam's multi_tensor_apply bursts; the k-th from the end is printed (default 3):
segments of continuous GPU activity (any stream) with their busy time, dominant kernels and the idle gap that follows."""
import csv, sys, glob, collections
f = glob.glob(sys.argv[1] + '/*/*kernel_trace.csv')[0]
k = int(sys.argv[2]) if len(sys.argv) > 2 else 3
rows = []
for r in csv.DictReader(open(f)):
    rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r.get('Stream_Id', r.get('Queue_Id', '?'))))
rows.sort()
def short(n):
    n = n.replace('(anonymous namespace)::', '').replace('void ', '').replace('at::native::', '')
    return n[:46]
# optimizer bursts: multi_tensor_apply kernels separated by < 2 ms belong to one step
opt = [s for s, e, n, q in rows if 'multi_tensor_apply' in n]
bursts = []
for s in opt:
    if not bursts or s - bursts[-1][-1] > 2_000_000: bursts.append([s])
    else: bursts[-1].append(s)
ends = [b[-1] for b in bursts if len(b) >= 4]          # Adam (several multi-tensor launches); the template SGD step is a 1-2 kernel burst
fused = [s for s, e, n, q in rows if 'adam_step_kernel' in n]          # the one-launch FusedAdam ends an iteration when it is in use
if fused:
    ends = fused
assert len(ends) > k + 1, len(ends)
t0, t1 = ends[-k - 2], ends[-k - 1]
it = [r for r in rows if t0 < r[0] <= t1 + 200_000]
print(f'iteration of {(t1 - t0) / 1e6:.2f} ms, {len(it)} kernels, streams {sorted(set(r[3] for r in it))}')
GAP = 25_000        # ns: a gap this long ends a segment
segs = []; cur = None; end = None
for s, e, n, q in it:
    if cur is None or s > end + GAP:
        if cur is not None: segs.append(cur)
        cur = dict(t=s, end=e, busy=0, names=collections.Counter(), n=0, last=s, streams=set())
        end = s
    cur['busy'] += max(0, e - max(s, end)); end = max(end, e); cur['end'] = end
    cur['names'][short(n)] += e - s; cur['n'] += 1; cur['streams'].add(q)
segs.append(cur)
tot_busy = sum(s['busy'] for s in segs)
print(f'busy {tot_busy / 1e6:.2f} ms in {len(segs)} segments; idle {(t1 - t0 - tot_busy) / 1e6:.2f} ms')
print(f'{"t ms":>7} {"len ms":>7} {"busy%":>5} {"#k":>4} {"gap after us":>12}  streams  top kernels (ms)')
for i, s in enumerate(segs):
    gap = (segs[i + 1]['t'] - s['end']) / 1e3 if i + 1 < len(segs) else 0
    ln = (s['end'] - s['t'])
    top = ', '.join(f'{n}={v / 1e6:.2f}' for n, v in s['names'].most_common(3))
    if ln < 150_000 and gap < 60: continue          # only the segments / gaps that matter
    print(f'{(s["t"] - t0) / 1e6:7.2f} {ln / 1e6:7.2f} {100 * s["busy"] / max(ln, 1):5.0f} {s["n"]:4d} {gap:12.1f}  {"".join(sorted(str(x)[-1] for x in s["streams"])):>7}  {top}')
# inside-segment idle (gaps < GAP) is host-launch-bound time
small = 0; end = None
for s, e, n, q in it:
    if end is not None and s > end and s - end <= GAP: small += s - end
    end = e if end is None else max(end, e)
print(f'idle in gaps <= {GAP / 1e3:.0f} us: {small / 1e6:.2f} ms; in longer gaps: {(t1 - t0 - tot_busy - small) / 1e6:.2f} ms')
