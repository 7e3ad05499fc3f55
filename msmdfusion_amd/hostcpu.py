"""Host-side placement of a rank's threads (no torch import: call it before anything
creates threads).

A training rank here is two busy Python threads (the step and the index prefetcher) plus
the HIP runtime's helpers.  On a 2-socket, 256-thread host the scheduler moves them between
cores and sockets from one time slice to the next, and the step time of the same binary on
the same box lands anywhere between 15.0 and 19.8 ms (100-133 samples/s over ten runs of
the LC line); restricted to a handful of CPUs -- any handful: 2 or 8 CPUs, either socket --
twelve runs in a row read 128.1-129.6 samples/s.  `pin_host_threads` gives
rank r the r-th group of `cpus_per_rank` CPUs of the process's affinity mask (sorted, which
lists physical cores before their SMT siblings on Linux).
"""
import os


def parse_cpu_list(spec):
    """'0-3,8,10-11' -> sorted list of ints"""
    out = set()
    for part in str(spec).split(","):
        part = part.strip()
        if "-" in part:
            a, b = part.split("-")
            out.update(range(int(a), int(b) + 1))
        elif part:
            out.add(int(part))
    return sorted(out)


def rank_cpus(local_rank, cpus_per_rank, allowed):
    """The r-th group of the allowed CPUs (wrapping when there are fewer groups than ranks)."""
    allowed = sorted(allowed)
    if len(allowed) <= cpus_per_rank:
        return allowed
    groups = len(allowed) // cpus_per_rank
    g = int(local_rank) % groups
    return allowed[g * cpus_per_rank:(g + 1) * cpus_per_rank]


def _ranges(cpus):
    out, run = [], []
    for c in sorted(cpus) + [None]:
        if run and (c is None or c != run[-1] + 1):
            out.append("%d-%d" % (run[0], run[-1]) if len(run) > 1 else str(run[0]))
            run = []
        if c is not None:
            run.append(c)
    return ",".join(out)


def base_cpus():
    """The CPUs the job may use: the affinity mask the first process of the job started with
    (kept in MSMD_AFFINITY_BASE so that child ranks spawned by an already pinned parent do not
    take the parent's four CPUs for the whole machine)."""
    if os.environ.get("MSMD_AFFINITY_BASE"):
        return parse_cpu_list(os.environ["MSMD_AFFINITY_BASE"])
    base = sorted(os.sched_getaffinity(0))
    os.environ["MSMD_AFFINITY_BASE"] = _ranges(base)
    return base


def quota_cpus():
    """CPUs' worth of time the cgroup grants (cpu.max quota / period), or None."""
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            return max(1, int(quota) // int(period))
    except (OSError, ValueError):
        pass
    return None


def within_quota(cpus=None):
    """As many CPUs of the job's mask as the cgroup's quota pays for.  A process confined to
    them cannot exceed the quota, whatever thread pools it starts -- which is what keeps the
    kernel from throttling it for the rest of every 100 ms period."""
    cpus = sorted(base_cpus() if cpus is None else cpus)
    q = quota_cpus()
    return cpus if q is None or q >= len(cpus) else cpus[:q]


def pin_host_threads(local_rank=None, cpus_per_rank=4):
    """Restrict the calling thread (and every thread it creates from now on) to this rank's
    CPUs.  MSMD_PIN=0 disables it, MSMD_PIN_CPUS='a-b,c' names the CPUs outright.
    -> the CPU list in effect, or None when pinning is off / unsupported."""
    if os.environ.get("MSMD_PIN", "1") != "1" or not hasattr(os, "sched_setaffinity"):
        return None
    if local_rank is None:
        local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    base = base_cpus()
    if os.environ.get("MSMD_PIN_CPUS"):
        cpus = parse_cpu_list(os.environ["MSMD_PIN_CPUS"])
    else:
        cpus = rank_cpus(local_rank, cpus_per_rank, base)
    try:
        os.sched_setaffinity(0, cpus)
    except OSError:
        return None
    return cpus


class unpinned:
    """with unpinned(): the calling thread may run (and create threads) on the job's CPUs
    again, as many as the cgroup's quota pays for -- for a CPU-side leg such as bench.py's
    cpu_baseline."""

    def __enter__(self):
        self.saved = None
        if hasattr(os, "sched_setaffinity"):
            self.saved = os.sched_getaffinity(0)
            try:
                os.sched_setaffinity(0, within_quota())
            except OSError:
                self.saved = None
        return self

    def __exit__(self, *exc):
        if self.saved is not None:
            os.sched_setaffinity(0, self.saved)
        return False
