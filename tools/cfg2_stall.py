#!/usr/bin/env python3
"""Where does the one slow step of the config-2 leg come from (23 ms inside 100 timed steps of 2.6 ms on the
driver's box)?  Replays bench.py's own leg ORDER in one fresh process -- model F fp16, model F bf16, then config 2 -- and stamps
every config-2 step: host time to ENQUEUE the step, HIP-event time of the step on the launch stream, the caching allocator's
hipMalloc / hipFree / retry counters, and every run of Python's cyclic garbage collector (generation, duration) -- then prints
the steps that took more than 1.3 x the median with everything that happened inside them.

    python tools/cfg2_stall.py [--steps 300] [--warmup 16] [--no-f-legs] [--reserve 0|1] [--graph 0|1] [--repeat 2]
"""
import argparse
import copy
import gc
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

ap = argparse.ArgumentParser()
ap.add_argument('--steps', type=int, default=300)
ap.add_argument('--warmup', type=int, default=16)
ap.add_argument('--no-f-legs', action='store_true')
ap.add_argument('--reserve', type=int, default=0, help='1: keypoints_amd.runtime.reserve_memory() after the first warm-up steps')
ap.add_argument('--graph', type=int, default=0)
ap.add_argument('--repeat', type=int, default=2)
ap.add_argument('--freeze', type=int, default=1)
mine = ap.parse_args()
sys.argv = sys.argv[:1] + ['--no-cpu-baseline', '--extra', '0']

import torch
import bench
import keypoints_amd as ka

dev = torch.device('cuda', 0)
torch.cuda.set_device(0)
base = bench.parse()
if not mine.no_f_legs:
    for prec in ('fp16', 'bf16'):
        f = copy.copy(base)
        f.precision, f.steps, f.warmup, f.profile_steps = prec, 20, 10, (3 if prec == 'fp16' else 0)
        r = bench.run_leg(f, 0, 1, dev)
        print(f'model F {prec}: {r["value"]:.1f} pairs/s  {r["ms_per_step"]:.3f} ms', flush=True)

from keypoints_amd.models import transporter

gc_log = []
_t_gc = [0.0]


def _gc_cb(phase, info):
    if phase == 'start':
        _t_gc[0] = time.perf_counter()
    else:
        gc_log.append((time.perf_counter(), info['generation'], 1e3 * (time.perf_counter() - _t_gc[0]), info.get('collected', 0)))


gc.callbacks.append(_gc_cb)
KEYS = ('num_device_alloc', 'num_device_free', 'num_alloc_retries', 'num_ooms', 'reserved_bytes.all.current', 'segment.all.allocated')


def memstat():
    s = torch.cuda.memory_stats()
    return {k: s.get(k, 0) for k in KEYS}


for rep in range(mine.repeat):
    torch.manual_seed(0)
    net = transporter.make('VGG_PONG_LAYERNECK', 1, 16, 4).to(dev).set_precision('bf16')
    opt = ka.FusedAdam(net.parameters(), lr=1e-4).bind(net)
    g = torch.Generator().manual_seed(1234)
    xs = (torch.rand(256, 1, 84, 84, generator=g) * 2 - 1).to(dev)
    xt = (torch.rand(256, 1, 84, 84, generator=g) * 2 - 1).to(dev)

    def step():
        opt.zero_grad()
        out = net(xs, xt)
        loss = ka.l2_reconstruction_loss(out[0], xt)
        loss.backward()
        opt.step()
        return loss

    if mine.graph:
        gstep = ka.graph.GraphedTrainStep(net, opt, None, (xs, xt), warmup=2)

        def step():   # noqa: F811
            return gstep(xs, xt)

    m0 = memstat()
    for i in range(mine.warmup):
        step()
        if i == 1 and mine.reserve:
            torch.cuda.synchronize()
            ka.runtime.reserve_memory()
    torch.cuda.synchronize()
    m1 = memstat()
    if mine.freeze:
        ka.runtime.freeze_gc()
    n = mine.steps
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    host, mem = [], []
    gc_log.clear()
    torch.cuda.synchronize()
    t00 = time.perf_counter()
    marks[0].record()
    for i in range(n):
        t0 = time.perf_counter()
        step()
        marks[i + 1].record()
        host.append((t0, time.perf_counter()))
        mem.append(memstat())
    torch.cuda.synchronize()
    wall = time.perf_counter() - t00
    ev = [marks[i].elapsed_time(marks[i + 1]) for i in range(n)]
    srt = sorted(ev)
    med = srt[n // 2]
    print(f'--- config 2, repeat {rep}: {256 * n / wall:.0f} pairs/s, mean {1e3 * wall / n:.3f} ms, median(event) {med:.3f}, max {srt[-1]:.3f}, '
          f'p99 {srt[int(0.99 * n)]:.3f}; host enqueue median {1e3 * sorted(b - a for a, b in host)[n // 2]:.3f} ms', flush=True)
    print(f'    allocator: warm-up {m0} -> {m1}; after timed {mem[-1]}')
    print(f'    gc runs in the timed region: {len(gc_log)} (gen2: {sum(1 for g_ in gc_log if g_[1] == 2)}), longest {max([g_[2] for g_ in gc_log], default=0):.2f} ms')
    # GPU-side wall clock of the step boundaries: mark 0 was recorded on an idle stream at t00
    tg = [t00]
    for i in range(n):
        tg.append(tg[-1] + 1e-3 * ev[i])

    def delta(i):
        prev = mem[i - 1] if i else m1
        return {k.split('.')[0]: mem[i][k] - prev[k] for k in KEYS if mem[i][k] != prev[k]}

    allocs = [(i, delta(i)) for i in range(n) if delta(i)]
    print(f'    host steps with allocator activity: {[(i, d) for i, d in allocs][:12]}')
    slow = [i for i in range(n) if ev[i] > 1.3 * med or 1e3 * (host[i][1] - host[i][0]) > 3 * med]
    for i in slow[:24]:
        a, b = host[i]
        gcs = [(gen, f'{ms:.2f} ms', col) for (t, gen, ms, col) in gc_log if a <= t <= b + 1e-3]
        line = f'    step {i:4d}: event {ev[i]:8.3f} ms  host enqueue {1e3 * (b - a):8.3f} ms  gc {gcs}  allocator {delta(i)}'
        if ev[i] > 2 * med and i > 0:
            # what the HOST was doing while the GPU sat in this step (the host runs ahead of the device)
            lo, hi = tg[i], tg[i + 1]
            hs = [j for j in range(n) if host[j][1] >= lo and host[j][0] <= hi]
            line += (f'\n        GPU interval = host steps {hs[0] if hs else None}..{hs[-1] if hs else None} (host ahead by {hs[0] - i if hs else 0} steps): '
                     f'enqueue ms {[round(1e3 * (host[j][1] - host[j][0]), 1) for j in hs][:12]}, allocator {[(j, delta(j)) for j in hs if delta(j)]}, '
                     f'gc {[(gen, round(ms, 2), col) for (t, gen, ms, col) in gc_log if lo - 5e-3 <= t <= hi]}')
        print(line)
    if not slow:
        print('    no step above 1.3 x median')
    if mine.freeze:
        gc.unfreeze()
    del net, opt, step
    gc.collect()
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
