"""Reads a stamp dump of tools/nt_lab.hip (one row per 128x128 tile: cycle stamps + hardware placement) and prints where a tile's
time goes: prologue / tile loop / epilogue durations, the gap between consecutive workgroups of one CU slot, the phase between the two
workgroups that share a CU, and the fraction of the launch in which a CU has 0 / 1 / 2 workgroups inside their tile loops."""
import csv, sys, collections, statistics as st


def main(path, ghz=None):
    rows = [{k: int(v) for k, v in r.items()} for r in csv.DictReader(open(path))]
    rows = [r for r in rows if r['t_end'] > 0]
    t0 = min(r['t_start'] for r in rows)
    span = max(r['t_end'] for r in rows) - t0
    for r in rows:
        hw = r['hw_id']
        r['wave'] = hw & 15; r['simd'] = (hw >> 4) & 3; r['cu'] = (hw >> 8) & 15; r['sh'] = (hw >> 12) & 1; r['se'] = (hw >> 13) & 7
        r['tg'] = (hw >> 16) & 15
        r['cukey'] = (r['xcc'] & 15, r['se'], r['sh'], r['cu'])
    pro = [r['t_pro'] - r['t_start'] for r in rows]; loop = [r['t_loop'] - r['t_pro'] for r in rows]; epi = [r['t_end'] - r['t_loop'] for r in rows]
    q = lambda v: (st.mean(v), sorted(v)[len(v) // 2], sorted(v)[int(len(v) * 0.9)])
    print(f"{len(rows)} tiles, launch span {span} ticks; distinct CUs {len(set(r['cukey'] for r in rows))}")
    for name, v in (('prologue', pro), ('tile loop', loop), ('epilogue', epi)):
        print(f"  {name:10s} mean {q(v)[0]:9.0f}  median {q(v)[1]:8d}  p90 {q(v)[2]:8d} ticks")
    by_cu = collections.defaultdict(list)
    for r in rows:
        by_cu[r['cukey']].append(r)
    gaps, both, one, none_, phase = [], 0, 0, 0, []
    ids = collections.Counter()
    for cu, rs in by_cu.items():
        rs.sort(key=lambda r: r['t_start'])
        ids.update((r['wave'], r['tg']) for r in rs)
        slots = collections.defaultdict(list)
        for r in rs:
            slots[r['tg']].append(r)
        for s, ss in slots.items():
            for a, b in zip(ss, ss[1:]):
                gaps.append(b['t_start'] - a['t_end'])
        # occupancy of the tile loop over time on this CU
        ev = []
        for r in rs:
            ev.append((r['t_pro'], 1)); ev.append((r['t_loop'], -1))
        ev.sort()
        lo, hi = min(r['t_start'] for r in rs), max(r['t_end'] for r in rs)
        cur, last = 0, lo
        for t, d in ev:
            if cur == 0: none_ += t - last
            elif cur == 1: one += t - last
            else: both += t - last
            cur += d; last = t
        none_ += hi - last
        if len(slots) == 2:
            k = sorted(slots)
            for a in slots[k[0]][1:-1]:
                nb = min(slots[k[1]], key=lambda b: abs(b['t_start'] - a['t_start']))
                phase.append(abs(nb['t_start'] - a['t_start']))
    tot = both + one + none_
    print(f"  (wave_id, tg_id) pairs seen: {dict(ids.most_common(8))}")
    if gaps:
        print(f"  gap between consecutive workgroups of one CU slot (tg_id): mean {st.mean(gaps):.0f} median {sorted(gaps)[len(gaps)//2]} ticks")
    if phase:
        print(f"  start-to-start phase between the two workgroups of a CU: median {sorted(phase)[len(phase)//2]} mean {st.mean(phase):.0f} ticks")
    print(f"  CU time with 2 / 1 / 0 workgroups inside the tile loop: {both/tot:.3f} / {one/tot:.3f} / {none_/tot:.3f}")
    if 'wall100mhz' in rows[0]:
        rate = [(r['t_end'] - r['t_start']) / (r['wall100mhz'] * 10.0) for r in rows if r['wall100mhz'] > 0]
        print(f"  s_memtime ticks per ns (against the 100 MHz s_memrealtime): median {sorted(rate)[len(rate)//2]:.3f}")
    tile = st.mean(r['t_end'] - r['t_start'] for r in rows)
    print(f"  mean tile residence {tile:.0f} ticks; loop share {st.mean(loop)/tile:.3f}")


if __name__ == '__main__':
    main(sys.argv[1])
