"""Host-core placement of the rank processes of one node (one process per GPU).

Each rank's launching thread enqueues ~1 100 kernels per step and runs only a few milliseconds ahead of its GPU; with 8 ranks
(plus their reader threads) free to float over every core of the machine, two launchers can land on one core or on the far
socket, and the step turns host-bound.  `pin_rank()` gives every rank of the node a DISJOINT slice of the allowed cores, taken
from the cores next to its GPU (the PCI device's `local_cpulist`) when sysfs shows them.  It must run BEFORE the first GPU call
of the process: threads that the HIP runtime and the data loader start later inherit the mask.

Nothing here touches the GPU: the GPU's PCI address comes from the KFD topology in sysfs (/sys/class/kfd), never from HIP.
The plan is a pure function of (allowed cores, per-GPU local cores, ranks on the node) so that every rank computes the same
one without talking to the others.  (The reference leaves placement to the OS: tools/scripts/once_train.sh:8-10 starts
torch.distributed.launch with 16 DataLoader workers per rank.)
"""
import glob
import os

__all__ = ['parse_cpulist', 'format_cpulist', 'gpu_local_cpus', 'plan_rank_cores', 'pin_rank', 'MIN_PIN_CORES']
MIN_PIN_CORES = 4


def parse_cpulist(text):
    """'0-3,8,10-11' -> [0, 1, 2, 3, 8, 10, 11]"""
    out = []
    for part in text.strip().split(','):
        if not part:
            continue
        if '-' in part:
            a, b = part.split('-')
            out.extend(range(int(a), int(b) + 1))
        else:
            out.append(int(part))
    return out


def format_cpulist(cpus):
    """[0, 1, 2, 3, 8, 10, 11] -> '0-3,8,10-11'"""
    cpus, parts, i = sorted(cpus), [], 0
    while i < len(cpus):
        j = i
        while j + 1 < len(cpus) and cpus[j + 1] == cpus[j] + 1:
            j += 1
        parts.append(str(cpus[i]) if i == j else f'{cpus[i]}-{cpus[j]}')
        i = j + 1
    return ','.join(parts)


def _core_order(cpus):
    """The cpus sorted so that SMT siblings sit next to each other (a slice then holds whole cores)."""
    key = {}
    for c in cpus:
        core, pkg = c, 0
        try:
            base = f'/sys/devices/system/cpu/cpu{c}/topology/'
            core = int(open(base + 'core_id').read())
            pkg = int(open(base + 'physical_package_id').read())
        except (OSError, ValueError):
            pass
        key[c] = (pkg, core, c)
    return sorted(cpus, key=lambda c: key[c])


def _kfd_gpu_nodes():
    """PCI addresses 'dddd:bb:dd.f' of the GPUs in KFD (= ROCr / HIP enumeration) order, [] when sysfs does not show them."""
    nodes = []
    try:
        dirs = sorted((p for p in glob.glob('/sys/class/kfd/kfd/topology/nodes/*') if os.path.basename(p).isdigit()),
                      key=lambda p: int(os.path.basename(p)))
        for d in dirs:
            try:
                props = dict(line.split() for line in open(os.path.join(d, 'properties')) if len(line.split()) == 2)
            except OSError:
                continue
            if int(props.get('simd_count', '0')) == 0:          # a CPU node
                continue
            loc, dom = int(props.get('location_id', '0')), int(props.get('domain', '0'))
            nodes.append('%04x:%02x:%02x.%x' % (dom, (loc >> 8) & 0xFF, (loc >> 3) & 0x1F, loc & 7))
    except (OSError, ValueError):                                # a sysfs this code does not understand: no locality, even split
        return []
    return nodes


def _visible(n):
    """Indices into the KFD GPU list that HIP will enumerate (ROCR_VISIBLE_DEVICES then HIP_VISIBLE_DEVICES), None if unparseable."""
    idx = list(range(n))
    for var in ('ROCR_VISIBLE_DEVICES', 'HIP_VISIBLE_DEVICES', 'CUDA_VISIBLE_DEVICES'):
        v = os.environ.get(var)
        if v is None or v == '':
            continue
        try:
            pick = [int(t) for t in v.split(',') if t.strip() != '']
        except ValueError:                                   # UUID form: give up on the mapping
            return None
        if any(p < 0 or p >= len(idx) for p in pick):
            return None
        idx = [idx[p] for p in pick]
    return idx


def gpu_local_cpus(device_index):
    """Cores next to HIP device `device_index` per sysfs (its PCI device's local_cpulist), or None."""
    nodes = _kfd_gpu_nodes()
    vis = _visible(len(nodes))
    if not nodes or vis is None or device_index >= len(vis):
        return None
    try:
        cpus = parse_cpulist(open(f'/sys/bus/pci/devices/{nodes[vis[device_index]]}/local_cpulist').read())
    except (OSError, ValueError):
        return None
    return cpus or None


def plan_rank_cores(allowed, local_cpus_per_rank, min_cores=2):
    """Disjoint core sets for the ranks of one node.

    allowed: the cores this job may use; local_cpus_per_rank[r]: the cores next to rank r's GPU, or None.
    Ranks whose GPUs share a locality (the same local core set within `allowed`) split that set evenly, in rank order, in
    whole-core order; a rank without locality information -- or whose local set cannot give every sharer `min_cores` cores --
    joins the pool that splits ALL allowed cores not claimed by a local group.  Returns one sorted list per rank; the lists are
    pairwise disjoint whenever `allowed` holds at least one core per rank (else every rank gets all of `allowed`)."""
    allowed = sorted(set(allowed))
    n = len(local_cpus_per_rank)
    if n == 0:
        return []
    if len(allowed) < n:
        return [list(allowed) for _ in range(n)]
    groups = {}
    for r, loc in enumerate(local_cpus_per_rank):
        key = tuple(sorted(set(loc) & set(allowed))) if loc else ()
        groups.setdefault(key, []).append(r)
    # a locality whose cores cannot feed its ranks falls back to the common pool
    pool_ranks = list(groups.pop((), []))
    for key in list(groups):
        if len(key) < min_cores * len(groups[key]):
            pool_ranks.extend(groups.pop(key))
    claimed = set(c for key in groups for c in key)
    out = [None] * n

    def deal(cores, ranks):
        cores = _core_order(list(cores))
        per = len(cores) // len(ranks)
        for j, r in enumerate(sorted(ranks)):
            out[r] = sorted(cores[j * per:(j + 1) * per])
    for key, ranks in groups.items():
        deal(key, ranks)
    if pool_ranks:
        rest = [c for c in allowed if c not in claimed]
        if len(rest) < len(pool_ranks):                      # the local groups took too much: split everything evenly instead
            deal(allowed, list(range(n)))
        else:
            deal(rest, pool_ranks)
    return out


def pin_rank(local_rank, local_world, device_indices=None, apply=True):
    """Pin this process (and every thread it starts afterwards) to its slice of the node's cores.  Call before the first GPU
    call.  device_indices[r] = the HIP device of local rank r (default: r) -- every rank must pass the same list, the plan is
    computed by each rank on its own.  Returns {'cores': [...], 'source': 'gpu-local' | 'even-split' | 'unpinned', 'allowed': n}.
    TMAE_PIN_CORES=0 turns the pinning off; with one rank on the node nothing is pinned unless TMAE_PIN_CORES=1.
    A slice below MIN_PIN_CORES (4; TMAE_PIN_MIN_CORES) is not applied: besides the launching thread a rank runs the HIP runtime's
    helper threads and RCCL's proxy thread, which polls -- all of them on one or two cores is worse than floating."""
    allowed = sorted(os.sched_getaffinity(0))
    env = os.environ.get('TMAE_PIN_CORES', '')
    if env == '0' or (local_world <= 1 and env != '1'):
        # one rank on the node: nothing to keep apart; the whole allowance stays (the launcher thread + HIP's helper threads)
        return {'cores': allowed, 'source': 'unpinned', 'allowed': len(allowed)}
    devs = list(device_indices) if device_indices is not None else list(range(local_world))
    locs = [gpu_local_cpus(d) for d in devs]
    mine = plan_rank_cores(allowed, locs)[local_rank]
    loc = locs[local_rank]
    source = 'gpu-local' if loc and set(mine) <= set(loc) else 'even-split'
    try:
        floor = int(os.environ.get('TMAE_PIN_MIN_CORES', str(MIN_PIN_CORES)))
    except ValueError:
        floor = MIN_PIN_CORES
    if len(mine) == len(allowed) or len(mine) < floor:
        return {'cores': allowed, 'source': 'unpinned', 'allowed': len(allowed)}
    if apply:
        try:
            os.sched_setaffinity(0, mine)
        except OSError:                                          # a cpuset that moved under us: stay where the OS put us
            return {'cores': allowed, 'source': 'unpinned', 'allowed': len(allowed)}
    return {'cores': mine, 'source': source, 'allowed': len(allowed)}
