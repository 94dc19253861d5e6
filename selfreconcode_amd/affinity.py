"""Where a rank's HOST threads run.

One iteration is ~2 000 C-ABI launches and ~600 torch launches issued by two threads that hand work to each other all the time (the
caller and the autograd engine's device thread) plus the HIP runtime's own.  On the GPU boxes of this pool (2 sockets, 256 logical
CPUs, mostly idle) the scheduler spreads those threads over the whole machine: every hand-over wakes a core out of an idle state, on
whatever socket.  Measured on the workload the host paces -- one frame per rank, configs[2]'s share of one of 8 GPUs,
`tools/host_profile.py` with SR_HP_FRAMES=1 SR_HP_SIM_WORLD=8, `profiles/r05_cpu_affinity.txt`: free 21.0-22.6 ms per step; confined
to one NUMA node 19.3-21.9; to 8 cores + their SMT siblings 18.5 with taskset, 19.5-21.1 through this module; to 2 logical CPUs 18.9; which node matters less than the confinement
(8 cores of the OTHER socket: 19.5).  At three frames per rank the GPU paces the step and placement does not show.

`bind()` confines the calling process -- every thread it has and every thread it creates later -- to `cores` physical cores (and their
SMT siblings) of the NUMA node its GPU hangs on.  Ranks that share a node take different groups, so 8 ranks on a 2 x 64-core box get 8
disjoint groups of 8: by default the group number is the position of the GPU among the GPUs of its node in PCI order, which every
process on the box computes alike -- single-GPU jobs of different tenants do not pile onto one group either.  `restore()` undoes it
(bench.py: before the CPU-baseline leg, which wants the whole machine).  Only ever narrows the set the process was started with
(taskset / cgroup limits are respected); does nothing where sysfs does not describe the machine.  SR_BIND_CPUS=0 switches it off,
SR_BIND_CORES=<n> sets the group size (default 8)."""
import os

_ORIGINAL = None
_SYS_CPU = "/sys/devices/system/cpu"
_SYS_NODE = "/sys/devices/system/node"


def parse_cpulist(text):
    """'0-3,8,10-11' -> [0, 1, 2, 3, 8, 10, 11]"""
    out = []
    for part in text.strip().split(","):
        if not part:
            continue
        a, _, b = part.partition("-")
        out.extend(range(int(a), int(b or a) + 1))
    return out


def _read(path):
    try:
        with open(path) as f:
            return f.read().strip()
    except OSError:
        return None


def gpu_pci_address(device_index):
    """'dddd:bb:dd.f' of a visible GPU, from the ids torch reports."""
    import torch
    p = torch.cuda.get_device_properties(device_index)
    return "%04x:%02x:%02x.0" % (int(getattr(p, "pci_domain_id", 0)), int(p.pci_bus_id), int(p.pci_device_id))


def gpu_numa_node(device_index):
    try:
        text = _read("/sys/bus/pci/devices/%s/numa_node" % gpu_pci_address(device_index))
        node = int(text) if text is not None else -1
    except (AttributeError, ValueError, RuntimeError, AssertionError):
        return None
    return node if node >= 0 else None


def gpus_by_node(drm_root="/sys/class/drm"):
    """{numa node: sorted PCI addresses of the GPUs on it} from the DRM cards sysfs lists -- ALL of the box's, whatever this process
    may see of them."""
    out = {}
    try:
        cards = os.listdir(drm_root)
    except OSError:
        return out
    for c in cards:
        if not c.startswith("card") or "-" in c:
            continue
        addr = os.path.basename(os.path.realpath(os.path.join(drm_root, c, "device")))
        if addr.count(":") != 2:                                   # (amdgpu_xcp_* partitions are not PCI functions)
            continue
        text = _read("/sys/bus/pci/devices/%s/numa_node" % addr)
        try:
            node = int(text)
        except (TypeError, ValueError):
            continue
        if node >= 0 and addr not in out.setdefault(node, []):
            out[node].append(addr)
    return {n: sorted(a) for n, a in out.items()}


def gpu_slot(device_index):
    """Position of this GPU among the GPUs of its NUMA node (PCI address order): the same answer in every process and every container
    on the box, so that neither the ranks of one job nor single-GPU jobs of different tenants pick the same core group."""
    try:
        addr = gpu_pci_address(device_index)
    except (AttributeError, RuntimeError, AssertionError):
        return None
    for node, addrs in gpus_by_node().items():
        if addr in addrs:
            return addrs.index(addr)
    return None


def node_cpus(node):
    text = _read("%s/node%d/cpulist" % (_SYS_NODE, node))
    return parse_cpulist(text) if text else None


def siblings_of(cpu):
    text = _read("%s/cpu%d/topology/thread_siblings_list" % (_SYS_CPU, cpu))
    return tuple(parse_cpulist(text)) if text else (cpu,)


def plan(allowed, preferred, siblings, slot, cores):
    """The logical CPUs of group `slot`: `cores` physical cores (each with all its SMT siblings, `siblings(cpu)`) taken in order from
    the allowed CPUs that are also in `preferred` (the GPU's node; None = no preference).  None when there is nothing to narrow -- fewer
    than two groups to choose from."""
    allowed = set(allowed)
    pool = allowed & set(preferred) if preferred else allowed
    if not pool:
        pool = allowed
    seen, phys = set(), []
    for c in sorted(pool):
        if c in seen:
            continue
        sib = tuple(s for s in siblings(c) if s in allowed) or (c,)
        seen.update(sib)
        phys.append(sib)
    groups = len(phys) // max(1, cores)
    if groups < 2:
        return None
    g = slot % groups
    return sorted(c for sib in phys[g * cores:(g + 1) * cores] for c in sib)


def _set_all_threads(cpus):
    """sched_setaffinity acts on ONE thread; the HIP runtime and torch have started theirs by the time the GPU is known."""
    me = os.getpid()
    try:
        tids = [int(t) for t in os.listdir("/proc/%d/task" % me)]
    except OSError:
        tids = [0]
    for t in tids:
        try:
            os.sched_setaffinity(t, cpus)
        except OSError:
            pass                                  # (a thread that exited meanwhile)
    os.sched_setaffinity(0, cpus)                 # the caller: what new threads inherit


def bind(device_index=None, slot=None, cores=None, fallback_slot=0):
    """`_bind`, but placement is an optimisation: whatever goes wrong (a sysfs layout this module has not seen, a sandbox that refuses
    sched_setaffinity) is reported in the record and the process runs where it was."""
    try:
        return _bind(device_index, slot, cores, fallback_slot)
    except Exception as e:          # noqa: BLE001 -- deliberately broad, see above
        try:
            restore()
        except Exception:           # noqa: BLE001
            pass
        return {"bound": False, "why": "%s: %s" % (type(e).__name__, e)}


def _bind(device_index=None, slot=None, cores=None, fallback_slot=0):
    """Confine this process as described above.  `slot` = which core group of the node; None: the position of the GPU among the GPUs of
    its node (`gpu_slot`), `fallback_slot` (the local rank) where sysfs cannot tell.  Returns a record of what was done (bench.py prints
    it) or of why nothing was."""
    global _ORIGINAL
    if os.environ.get("SR_BIND_CPUS", "1") == "0":
        return {"bound": False, "why": "SR_BIND_CPUS=0"}
    if not hasattr(os, "sched_setaffinity"):
        return {"bound": False, "why": "no sched_setaffinity on this platform"}
    cores = int(os.environ.get("SR_BIND_CORES", "8")) if cores is None else int(cores)
    allowed = sorted(os.sched_getaffinity(0)) if _ORIGINAL is None else sorted(_ORIGINAL)
    node = gpu_numa_node(device_index) if device_index is not None else None
    slot_from = "given"
    if slot is None:
        slot = gpu_slot(device_index) if device_index is not None else None
        slot_from = "position of the GPU on its NUMA node"
        if slot is None:
            slot, slot_from = fallback_slot, "local rank"
    preferred = node_cpus(node) if node is not None else None
    chosen = plan(allowed, preferred, siblings_of, slot, cores)
    if not chosen:
        return {"bound": False, "why": "%d allowed CPUs: nothing to narrow" % len(allowed), "numa_node": node}
    if _ORIGINAL is None:
        _ORIGINAL = set(allowed)
    _set_all_threads(chosen)
    return {"bound": True, "cpus": _compact(chosen), "logical_cpus": len(chosen), "physical_cores": cores, "numa_node": node, "slot": slot,
            "slot_from": slot_from, "of_allowed": len(allowed)}


def restore():
    """Back to the set the process was started with."""
    global _ORIGINAL
    original, _ORIGINAL = _ORIGINAL, None
    if original is not None:
        _set_all_threads(original)


def _compact(cpus):
    """[0,1,2,3,8] -> '0-3,8'"""
    out, run = [], []
    for c in list(cpus) + [None]:
        if run and (c is None or c != run[-1] + 1):
            out.append(str(run[0]) if len(run) == 1 else "%d-%d" % (run[0], run[-1]))
            run = []
        if c is not None:
            run.append(c)
    return ",".join(out)
