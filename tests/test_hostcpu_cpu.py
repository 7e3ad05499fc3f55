"""msmdfusion_amd.hostcpu: which CPUs a rank's threads are pinned to."""
import os
import subprocess
import sys

from msmdfusion_amd import hostcpu as H

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cpu_list_and_rank_groups():
    assert H.parse_cpu_list("0-3,8,10-11") == [0, 1, 2, 3, 8, 10, 11]
    assert H._ranges([0, 1, 2, 3, 8, 10, 11]) == "0-3,8,10-11"
    allowed = list(range(0, 64)) + list(range(128, 192))
    groups = [H.rank_cpus(r, 4, allowed) for r in range(8)]
    assert groups[0] == [0, 1, 2, 3] and groups[7] == [28, 29, 30, 31]
    assert len({c for g in groups for c in g}) == 32              # disjoint
    assert H.rank_cpus(5, 4, [7, 3]) == [3, 7]                     # fewer CPUs than a group
    assert H.rank_cpus(3, 2, [0, 1, 2, 3]) == [2, 3] and H.rank_cpus(2, 2, [0, 1, 2, 3]) == [0, 1]


def test_pinning_in_a_child_process_and_its_children():
    """A pinned parent hands its ORIGINAL mask on (MSMD_AFFINITY_BASE): a rank spawned from
    it picks its group from the whole job's CPUs, not from the parent's four."""
    child = ("import os, sys; sys.path.insert(0, os.environ['MSMD_TEST_ROOT']); "
             "from msmdfusion_amd import hostcpu as H; "
             "print(H.pin_host_threads(local_rank=1, cpus_per_rank=1))")
    code = (
        "import os, sys, subprocess\n"
        "sys.path.insert(0, os.environ['MSMD_TEST_ROOT'])\n"
        "from msmdfusion_amd import hostcpu as H\n"
        "before = sorted(os.sched_getaffinity(0))\n"
        "mine = H.pin_host_threads(local_rank=0, cpus_per_rank=1)\n"
        "assert sorted(os.sched_getaffinity(0)) == mine == before[:1], (mine, before)\n"
        "with H.unpinned():\n"
        "    assert sorted(os.sched_getaffinity(0)) == before\n"
        "assert sorted(os.sched_getaffinity(0)) == mine\n"
        "child = subprocess.check_output([sys.executable, '-c', %r], env=os.environ)\n"
        "print(before, child.decode().strip())\n" % child)
    env = {k: v for k, v in os.environ.items() if not k.startswith("MSMD_")}
    env["MSMD_TEST_ROOT"] = ROOT
    out = subprocess.check_output([sys.executable, "-c", code], env=env).decode().split("] ")
    before = eval(out[0] + "]")
    if len(before) > 1:
        assert eval(out[1]) == [before[1]]
    off = subprocess.check_output(
        [sys.executable, "-c", "import os, sys; sys.path.insert(0, os.environ['MSMD_TEST_ROOT']); "
         "from msmdfusion_amd import hostcpu as H; print(H.pin_host_threads())"],
        env=dict(env, MSMD_PIN="0"))
    assert off.decode().strip() == "None"


def test_eight_ranks_fit_a_sixteen_cpu_quota(monkeypatch):
    """SCALE runs 8 ranks on a box whose cgroup pays for 16 CPUs: the ranks' CPU groups are
    disjoint, equal, and together inside the quota (round 2 gave every rank 4 = 32 CPUs);
    with two busy threads per rank the host counts as oversubscribed only beyond the quota."""
    base = list(range(0, 64)) + list(range(128, 192))
    monkeypatch.setenv("MSMD_CPU_QUOTA", "16")
    groups = [H.plan_rank_cpus(r, 8, base) for r in range(8)]
    assert all(len(g) == 2 for g in groups)
    flat = [c for g in groups for c in g]
    assert len(set(flat)) == 16 and set(flat) == set(base[:16])
    assert [H.plan_rank_cpus(r, 4, base) for r in range(4)] == \
        [base[4 * r:4 * r + 4] for r in range(4)]
    assert H.plan_rank_cpus(0, 1, base) == base[:4]
    assert H.plan_rank_cpus(5, 32, base) == [base[5]]            # never less than one CPU
    # 3 busy threads per rank in a multi-rank job (step, index prefetcher, RCCL proxy)
    assert not H.host_is_oversubscribed(1) and not H.host_is_oversubscribed(5)
    assert H.host_is_oversubscribed(6) and H.host_is_oversubscribed(8)
    monkeypatch.setenv("MSMD_CPU_QUOTA", "8")
    assert H.host_is_oversubscribed(4) and not H.host_is_oversubscribed(2)
    assert [len(H.plan_rank_cpus(r, 8, base)) for r in range(8)] == [1] * 8
    monkeypatch.delenv("MSMD_CPU_QUOTA")
    monkeypatch.setenv("LOCAL_WORLD_SIZE", "8")
    assert H.local_world_size() == 8
