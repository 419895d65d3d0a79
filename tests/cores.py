"""Host cores the CPU oracle / emulator tests may really use."""
import os


def usable_cores() -> int:
    """Cores this process may really use (affinity mask, capped by the cgroup quota and 32)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()[:2]
        if quota != 'max':
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, min(n, 32))
