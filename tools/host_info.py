"""What the GPU box gives a process: CPU model, logical CPUs, affinity, cgroup CPU quota, memory."""
import os
print("os.cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/memory.max", "/sys/fs/cgroup/pids.max"):
    try:
        print(f, open(f).read().strip())
    except OSError as e:
        print(f, "n/a", e.__class__.__name__)
os.system("lscpu | grep -E 'Model name|^CPU\\(s\\)|Thread|Socket|NUMA node\\(s\\)'; nproc; free -g | head -2; uptime")
