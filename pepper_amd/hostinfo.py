"""How many CPUs this process may really use: the scheduler affinity AND the cgroup CPU quota.

`os.cpu_count()` reports the host's logical CPUs; a container with `cpu.max = 1600000 100000` (the MI355X boxes of this
project: 256 logical CPUs visible, 16 CPUs' worth of time) runs 64 busy processes four times slower each.  Everything that
sizes a pool of host workers -- reader / writer lanes, the CPU baselines of bench.py -- goes through usable_cpus()."""
import math
import os


def cgroup_cpu_quota():
    """CPUs' worth of time the cgroup grants (float), or None when unlimited / unknown."""
    try:
        with open("/sys/fs/cgroup/cpu.max") as fh:                      # cgroup v2
            quota, period = fh.read().split()[:2]
        if quota != "max":
            return float(quota) / float(period)
        return None
    except (OSError, ValueError):
        pass
    try:
        with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as fh:         # cgroup v1
            quota = int(fh.read())
        with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as fh:
            period = int(fh.read())
        if quota > 0 and period > 0:
            return quota / period
    except (OSError, ValueError):
        pass
    return None


def usable_cpus():
    try:
        n = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        n = os.cpu_count() or 1
    quota = cgroup_cpu_quota()
    if quota is not None:
        n = min(n, max(1, int(math.floor(quota + 1e-9))))
    return max(1, n)
