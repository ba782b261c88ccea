cat /sys/fs/cgroup/cpu.max 2>/dev/null; cat /sys/fs/cgroup/cpu/cpu.cfs_quota_us /sys/fs/cgroup/cpu/cpu.cfs_period_us 2>/dev/null
nproc; python3 - <<'PY'
import os, time, multiprocessing as mp
print("affinity", len(os.sched_getaffinity(0)))
def burn(_):
    t0 = time.time(); x = 0
    while time.time() - t0 < 2.0:
        for i in range(10000): x += i * i
    return x
for n in (1, 8, 32, 64, 128, 256):
    t0 = time.time(); c0 = os.times()
    with mp.Pool(n) as p: p.map(burn, range(n))
    c1 = os.times(); dt = time.time() - t0
    print(n, "procs: wall", round(dt, 2), "child cpu", round((c1.children_user - c0.children_user), 1), "=> parallelism", round((c1.children_user - c0.children_user) / dt, 1))
PY
grep -c processor /proc/cpuinfo; grep "model name" /proc/cpuinfo | head -1; cat /proc/loadavg
