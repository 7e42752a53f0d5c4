"""Host-side environment of the launch thread.

A training step of this path is ~400 kernel launches that ONE host thread enqueues in ~6 ms; the GPU needs ~35 ms for them.
That only works while nothing else in the process burns the container's CPU budget: on the GPU boxes the container sees 256
cores but has a cgroup quota of 16 CPUs (`/sys/fs/cgroup/cpu.max` = `1600000 100000`).  torch sizes its OpenMP pool by the
core count, so ONE torch CPU reduction per step (e.g. `mask.all(-1)` on the host copy of a padding mask) wakes 256 spinning
threads, the cgroup's quota for the 100 ms period is gone within ~10 ms, and the kernel throttles EVERY thread of the
container -- the launch thread included -- for the remaining ~90 ms.  Measured on the same box, same build
(profiles/r04/host_stalls.txt): 52 ms per step with 41 launch-thread stalls of ~90 ms in 80 steps, against 34.5 ms per step
(= the GPU-side step time) with the pool capped.

The package's own per-step host work is numpy / plain Python (single-threaded); `cap_threads()` is for the process around
it: bench.py calls it, a training script should ("Keeping the launch thread ahead of the GPU").
"""
import math
import os


def cpu_quota():
    """CPUs the container may use per scheduling period (cgroup v2 cpu.max, v1 cfs quota), or None if unlimited / unknown"""
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, p = f.read().split()[:2]
        if q != "max":
            return float(q) / float(p)
        return None
    except (OSError, ValueError):
        pass
    try:
        with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f:
            q = float(f.read())
        with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
            p = float(f.read())
        return q / p if q > 0 else None
    except (OSError, ValueError):
        return None


def usable_cpus():
    """min(cores in the affinity mask, cgroup quota), at least 1"""
    try:
        n = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        n = os.cpu_count() or 1
    q = cpu_quota()
    if q is not None:
        n = min(n, max(1, int(math.floor(q))))
    return max(1, n)


def cap_threads(limit=4):
    """torch's intra-op pool (OpenMP) and inter-op pool down to min(limit, usable_cpus()); returns the new intra-op count.
    limit=None: only the container's own budget."""
    import torch
    n = usable_cpus() if limit is None else max(1, min(int(limit), usable_cpus()))
    if torch.get_num_threads() > n:
        torch.set_num_threads(n)
    return torch.get_num_threads()
