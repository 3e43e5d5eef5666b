"""Host-core placement of the ranks of one node (tmae_amd/train/affinity.py): the plan is disjoint across
ranks for the topologies an 8-GPU MI355X node can show, and two real processes pinning themselves end up on disjoint cores."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

sys.path.insert(0, os.path.join(ROOT, 't-mae_amd'))
AFF = os.path.join(ROOT, 't-mae_amd', 'tmae_amd', 'train', 'affinity.py')


def _mod():
    import importlib.util
    spec = importlib.util.spec_from_file_location('_aff_t', AFF)
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def _disjoint(plan):
    return all(not (set(a) & set(b)) for i, a in enumerate(plan) for b in plan[i + 1:])


def test_cpulist_round_trip():
    m = _mod()
    assert m.parse_cpulist('0-3,8,10-11\n') == [0, 1, 2, 3, 8, 10, 11]
    assert m.format_cpulist([11, 0, 1, 2, 3, 8, 10]) == '0-3,8,10-11'
    assert m.parse_cpulist('') == [] and m.format_cpulist([]) == ''


@pytest.mark.parametrize('allowed,locs', [
    (range(256), [list(range(0, 64)) + list(range(128, 192))] * 4 + [list(range(64, 128)) + list(range(192, 256))] * 4),   # 2 sockets, SMT
    (range(128), [list(range(16 * (r // 1), 16 * (r // 1) + 16)) for r in range(8)]),                                   # NPS4-like: a domain per GPU
    (range(128), [None] * 8),                                                                                           # no sysfs
    (range(32), [list(range(0, 128))] * 8),                                                                             # cgroup narrower than the locality
    (range(16), [list(range(0, 8)), list(range(8, 16))]),
    ([0, 1, 2, 3, 4, 5, 6, 7], [list(range(0, 4)), None]),                                                              # mixed: one rank without locality
    (range(96), [list(range(0, 48))] * 6 + [list(range(48, 96))] * 2),                                                 # uneven GPU counts per socket
    (range(64), [list(range(0, 2))] * 2 + [None] * 2),                                                                  # a locality too small for its ranks
])
def test_plan_is_disjoint_and_within_allowed(allowed, locs):
    m = _mod()
    plan = m.plan_rank_cores(allowed, locs)
    assert len(plan) == len(locs)
    assert all(len(p) >= 1 for p in plan)
    assert all(set(p) <= set(allowed) for p in plan)
    assert _disjoint(plan), plan
    # a rank whose GPU-local cores can feed all the ranks that share them stays inside them
    for r, loc in enumerate(locs):
        if loc:
            sharers = sum(1 for o in locs if o and set(o) & set(allowed) == set(loc) & set(allowed))
            if len(set(loc) & set(allowed)) >= 2 * sharers:
                assert set(plan[r]) <= set(loc), (r, plan[r])


def test_fewer_cores_than_ranks_is_not_pinned():
    m = _mod()
    assert m.plan_rank_cores([0, 1], [None] * 4) == [[0, 1]] * 4


_CHILD = r'''
import importlib.util, json, os, sys
spec = importlib.util.spec_from_file_location('_aff_c', sys.argv[1]); m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
pin = m.pin_rank(int(os.environ['LOCAL_RANK']), int(os.environ['LOCAL_WORLD_SIZE']), device_indices=[0, 0])
import threading
seen = []
t = threading.Thread(target=lambda: seen.append(sorted(os.sched_getaffinity(0))))      # a thread started AFTER the pin inherits it
t.start(); t.join()
print(json.dumps({'rank': int(os.environ['LOCAL_RANK']), 'plan': pin['cores'], 'main': sorted(os.sched_getaffinity(0)), 'thread': seen[0],
                  'source': pin['source']}))
'''


def test_two_ranks_pin_themselves_to_disjoint_cores():
    """The 2-rank rehearsal of bench.py / tools/train.py: both processes compute the plan on their own (no communication), apply it
    before anything else, and their masks -- and those of threads they start afterwards -- do not intersect."""
    allowed = sorted(os.sched_getaffinity(0))
    if len(allowed) < 2:
        pytest.skip('one core')
    outs = []
    for r in range(2):
        env = dict(os.environ, LOCAL_RANK=str(r), LOCAL_WORLD_SIZE='2', TMAE_PIN_MIN_CORES='1')
        env.pop('TMAE_PIN_CORES', None)
        res = subprocess.run([sys.executable, '-c', _CHILD, AFF], env=env, capture_output=True, text=True, timeout=60)
        assert res.returncode == 0, res.stderr
        outs.append(json.loads(res.stdout.strip().splitlines()[-1]))
    print('rank cores:', [o['main'] for o in outs])
    for o in outs:
        assert o['main'] == o['plan'] == o['thread'] and set(o['main']) <= set(allowed) and o['source'] != 'unpinned'
    assert not (set(outs[0]['main']) & set(outs[1]['main']))
    assert len(outs[0]['main']) + len(outs[1]['main']) >= len(allowed) - 1


def test_single_rank_keeps_its_allowance():
    m = _mod()
    before = sorted(os.sched_getaffinity(0))
    pin = m.pin_rank(0, 1)
    assert pin['source'] == 'unpinned' and pin['cores'] == before and sorted(os.sched_getaffinity(0)) == before


def test_a_slice_below_the_floor_is_not_applied(monkeypatch):
    """8 ranks on a node that allows this job 8 cores would get one core each -- launcher, HIP helper threads and RCCL's polling
    proxy thread on one core: the plan is then reported as unpinned and nothing is applied (MIN_PIN_CORES, TMAE_PIN_MIN_CORES)."""
    m = _mod()
    before = sorted(os.sched_getaffinity(0))
    monkeypatch.setenv('TMAE_PIN_MIN_CORES', str(len(before) + 1))
    monkeypatch.delenv('TMAE_PIN_CORES', raising=False)
    pin = m.pin_rank(0, 2, device_indices=[0, 0])
    assert pin['source'] == 'unpinned' and pin['cores'] == before and sorted(os.sched_getaffinity(0)) == before
    assert m.MIN_PIN_CORES >= 2
