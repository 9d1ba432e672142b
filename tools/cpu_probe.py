import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import kp_oracle as O
print('cpu_count', os.cpu_count(), 'affinity', len(os.sched_getaffinity(0)))
for f in ('/sys/fs/cgroup/cpu.max', '/sys/fs/cgroup/cpu/cpu.cfs_quota_us'):
    if os.path.exists(f): print(f, open(f).read().strip())
plans = O.transporter_plans('F', 3, 64, 10)
P, Bf = O.init_params(plans, seed=0)
g = torch.Generator().manual_seed(1)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
xs = torch.rand(B, 3, 128, 128, generator=g); xt = torch.rand(B, 3, 128, 128, generator=g)
for nt in (8, 16, 32, 64):
    torch.set_num_threads(nt)
    st = {}
    t = time.time(); O.transporter_train_step(plans, P, Bf, st, xs, xt); t1 = time.time() - t
    t = time.time(); O.transporter_train_step(plans, P, Bf, st, xs, xt); t2 = time.time() - t
    print(nt, 'threads: first %.2fs second %.2fs -> %.2f pairs/s' % (t1, t2, B / t2), flush=True)
