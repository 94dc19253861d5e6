"""How much does the step depend on the host?  Runs bench.py with a busy-wait of `us` microseconds in front of every C-ABI call of the
package (~170 calls + the refiner's chains per iteration at one frame, ~2 000 launches' worth at three):

    python tools/slow_host.py <us> [bench.py arguments]        e.g.  python tools/slow_host.py 4 --steps 20 --warmup 5 --no-fine --no-extra-records --no-cpu-baseline

Round 5 (profiles/r05_summary.md): +2 us per call costs 0.1 ms per iteration, +4 us 0.9 ms."""
import os
import runpy
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from selfreconcode_amd import _lib  # noqa: E402

us = float(sys.argv[1])
_call = _lib.call


def slow(name, *a):
    t = time.perf_counter()
    while (time.perf_counter() - t) * 1e6 < us:
        pass
    return _call(name, *a)


_lib.call = slow
sys.argv = [os.path.join(ROOT, 'bench.py')] + sys.argv[2:]
runpy.run_path(os.path.join(ROOT, 'bench.py'), run_name='__main__')
