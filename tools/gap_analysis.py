"""GPU idle-gap analysis of a rocprofv3 --kernel-trace CSV: python tools/gap_analysis.py <output dir> [first_iteration last_iteration+1].
Reports busy / idle time and which kernel transitions the idle time sits between (host-side launch cost, syncs)."""
import csv, sys, glob, collections
f = glob.glob(sys.argv[1] + '/*/*kernel_trace.csv')[0]
rows = []
for r in csv.DictReader(open(f)):
    rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']))
rows.sort()
if len(sys.argv) > 3:
    # window of whole iterations [A, B) counted in Adam steps: the one-launch FusedAdam kernel (adam_step_kernel) ends an iteration; with
    # torch.optim.Adam (bench.py --torch-adam) a burst of >= 4 multi_tensor_apply kernels does
    A, B = int(sys.argv[2]), int(sys.argv[3])
    ends = [s for s, e, n in rows if 'adam_step_kernel' in n]        # (one launch per step for <= 64 parameter tensors)
    if not ends:
        bursts = []
        for s, e, n in rows:
            if 'multi_tensor_apply' in n:
                if not bursts or s - bursts[-1][-1] > 2_000_000: bursts.append([s])
                else: bursts[-1].append(s)
        ends = [b[-1] for b in bursts if len(b) >= 4]
    t0, t1 = ends[A - 1], ends[B - 1]
    rows = [r for r in rows if t0 < r[0] <= t1 + 100_000]
    print(f'window: iterations {A}..{B - 1} of {len(ends)} ({(t1 - t0) / 1e6 / (B - A):.2f} ms / iteration under tracing)')
else:
    # keep the last 60% (steady state)
    n0 = int(len(rows) * 0.4)
    rows = rows[n0:]
span = rows[-1][1] - rows[0][0]
busy = 0; gaps = collections.Counter(); gapn = collections.Counter(); end = rows[0][0]
big = []
def short(n):
    n = n.replace('(anonymous namespace)::', '').replace('void ', '')
    return n[:60]
prev = 'start'
hist = collections.Counter()
for s, e, n in rows:
    if s > end:
        g = s - end
        key = short(prev) + '  ->  ' + short(n)
        gaps[key] += g; gapn[key] += 1
        hist[min(int(g / 5000), 20)] += g
        big.append((g, key))
        busy += e - s
    else:
        busy += max(0, e - max(s, end))
    if e > end: end = e; prev = n
print(f'kernels {len(rows)}  span {span/1e6:.1f} ms  busy {busy/1e6:.1f} ms  idle {(span-busy)/1e6:.1f} ms ({100*(span-busy)/span:.1f}%)')
print('idle by gap size (5us bins, last = >=100us):', {k*5: round(v/1e6, 2) for k, v in sorted(hist.items())})
print('top gap transitions (total ms, count, avg us):')
for k, v in gaps.most_common(40):
    print(f'  {v/1e6:7.2f} ms  {gapn[k]:5d}  {v/gapn[k]/1e3:7.1f} us   {k}')
big.sort(reverse=True)
print('largest single gaps:')
for g, k in big[:25]: print(f'  {g/1e3:8.1f} us  {k}')
