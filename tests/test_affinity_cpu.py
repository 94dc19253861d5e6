"""Host-thread placement (selfreconcode_amd/affinity.py): the planning is pure and is tested on a description of the GPU boxes'
topology (2 sockets x 64 cores x 2 threads, the numbering `lscpu` shows there); bind / restore run for real in a child process."""
import os
import subprocess
import sys

import pytest

from selfreconcode_amd import affinity as A

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BOX = list(range(256))
NODE = {0: A.parse_cpulist("0-63,128-191"), 1: A.parse_cpulist("64-127,192-255")}
smt = lambda c: (c % 128, c % 128 + 128)


def test_cpulists_round_trip():
    assert A.parse_cpulist("0-3,8,10-11\n") == [0, 1, 2, 3, 8, 10, 11]
    assert A._compact([0, 1, 2, 3, 8, 10, 11]) == "0-3,8,10-11"
    assert A.parse_cpulist(A._compact(NODE[1])) == NODE[1]
    assert A.parse_cpulist("") == []


def test_eight_ranks_get_disjoint_core_groups_on_their_own_node():
    taken = set()
    for local_rank in range(8):
        node = 0 if local_rank < 4 else 1                    # four GPUs per socket
        cpus = A.plan(BOX, NODE[node], smt, local_rank, 8)
        assert len(cpus) == 16 and set(cpus) <= set(NODE[node])
        assert all(smt(c)[0] in cpus and smt(c)[1] in cpus for c in cpus)        # whole cores: both threads of each
        assert not (taken & set(cpus))
        taken |= set(cpus)
    assert A.plan(BOX, NODE[1], smt, 8, 8) == A.plan(BOX, NODE[1], smt, 0, 8)      # more ranks than groups: wraps around


def test_the_plan_only_narrows_what_the_process_was_given():
    given = list(range(64, 96))                              # e.g. a cgroup of 32 CPUs on node 1, no SMT siblings among them
    cpus = A.plan(given, NODE[1], smt, 1, 8)
    assert cpus == list(range(72, 80))
    assert A.plan(given, NODE[0], smt, 0, 8) == list(range(64, 72))      # nothing allowed on the GPU's node: any allowed core group
    assert A.plan(list(range(8)), None, smt, 0, 8) is None                # fewer than two groups: left alone
    assert A.plan(list(range(15)), None, lambda c: (c,), 3, 8) is None


CHILD = r'''
import os, sys, threading, time
sys.path.insert(0, %r)
from selfreconcode_amd import affinity as A
stop = threading.Event()
t = threading.Thread(target=stop.wait); t.start()            # a thread that exists BEFORE the bind (the HIP runtime's, torch's pools)
before = sorted(os.sched_getaffinity(0))
rec = A.bind(None, slot=0, cores=1)
tids = [int(x) for x in os.listdir('/proc/self/task')]
sets = {tuple(sorted(os.sched_getaffinity(x))) for x in tids}
born = []
u = threading.Thread(target=lambda: born.append(sorted(os.sched_getaffinity(0)))); u.start(); u.join()   # a thread created AFTER it
A.restore()
after = {tuple(sorted(os.sched_getaffinity(x))) for x in tids}
stop.set(); t.join()
print(repr((before, rec, sorted(sets), born, sorted(after))))
''' % ROOT


@pytest.mark.skipif(not hasattr(os, "sched_setaffinity"), reason="Linux only")
def test_bind_moves_every_thread_and_restore_undoes_it():
    env = dict(os.environ); env.pop("SR_BIND_CPUS", None)
    out = subprocess.run([sys.executable, "-c", CHILD], capture_output=True, text=True, env=env, timeout=120)
    assert out.returncode == 0, out.stderr[-2000:]
    before, rec, during, born, after = eval(out.stdout.strip().splitlines()[-1])
    if len(before) < 2:
        assert rec["bound"] is False
        return
    assert rec["bound"] and rec["of_allowed"] == len(before)
    chosen = A.parse_cpulist(rec["cpus"])
    assert set(chosen) < set(before)
    assert during == [tuple(chosen)] and born == [chosen]                   # the old thread, the caller and the new thread alike
    assert after == [tuple(before)]
    off = subprocess.run([sys.executable, "-c", CHILD], capture_output=True, text=True, env=dict(env, SR_BIND_CPUS="0"), timeout=120)
    assert eval(off.stdout.strip().splitlines()[-1])[1] == {"bound": False, "why": "SR_BIND_CPUS=0"}


def test_gpu_position_on_its_node_from_a_sysfs_tree(tmp_path, monkeypatch):
    """The listing of a pool box: eight GPUs (PCI functions) among 64 DRM cards, the rest compute partitions without a PCI address."""
    pci = {"0000:0a:00.0": 0, "0000:2e:00.0": 0, "0000:54:00.0": 0, "0000:72:00.0": 0, "0000:8b:00.0": 1, "0000:a4:00.0": 1, "0000:bd:00.0": 1, "0000:d9:00.0": 1}
    drm = tmp_path / "drm"; drm.mkdir()
    devs = tmp_path / "devices"; devs.mkdir()
    for i, addr in enumerate(sorted(pci, reverse=True)):                  # card numbers do not follow the PCI order
        (devs / addr).mkdir()
        (drm / f"card{8 * i}").mkdir()
        os.symlink(devs / addr, drm / f"card{8 * i}" / "device")
        (drm / f"card{8 * i}-DP-1").mkdir()                               # connector entries are skipped
        for j in range(1, 8):
            x = tmp_path / f"amdgpu_xcp_{7 * i + j}"; x.mkdir()
            (drm / f"card{8 * i + j}").mkdir()
            os.symlink(x, drm / f"card{8 * i + j}" / "device")
    monkeypatch.setattr(A, "_read", lambda path: str(pci[os.path.basename(os.path.dirname(path))]) if path.endswith("numa_node") else None)
    by_node = A.gpus_by_node(str(drm))
    assert by_node == {0: sorted(a for a in pci if pci[a] == 0), 1: sorted(a for a in pci if pci[a] == 1)}
    monkeypatch.setattr(A, "gpus_by_node", lambda: by_node)
    for addr, want in (("0000:0a:00.0", 0), ("0000:72:00.0", 3), ("0000:8b:00.0", 0), ("0000:d9:00.0", 3)):
        monkeypatch.setattr(A, "gpu_pci_address", lambda i, a=addr: a)
        assert A.gpu_slot(0) == want
    monkeypatch.setattr(A, "gpu_pci_address", lambda i: "0000:ff:00.0")
    assert A.gpu_slot(0) is None                                           # unknown device: the caller falls back to the local rank


def test_a_refused_placement_is_reported_not_raised(monkeypatch):
    def refuse(*a):
        raise PermissionError("sandbox")
    monkeypatch.setattr(A, "_set_all_threads", refuse)
    monkeypatch.delenv("SR_BIND_CPUS", raising=False)
    rec = A.bind(None, slot=0, cores=1)
    assert rec["bound"] is False and ("PermissionError" in rec["why"] or "nothing to narrow" in rec["why"])
    assert A._ORIGINAL is None
