#!/usr/bin/env python3
"""Concurrency report from a rocprofv3 kernel trace (`--kernel-trace --output-format csv`): how much of the wall time has
matrix-core kernels running, HBM-bound passes running, both at once, or nothing.
python tools/overlap_report.py <..._kernel_trace.csv> [skip_fraction]"""
import csv, sys, collections

MFMA = ('conv3x3_patch_k', 'wgrad3x3_patch_k', 'conv3x3_mfma_k', 'wgrad3x3_mfma_k', 'conv3x3_f32', 'wgrad3x3_f32')


def short_name(name):
    return name.replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0]


def hbm_only_report(ev, lo, hi, steps):
    """Which kernels run while NO matrix-core kernel does (the time overlap does not hide), and where in the step."""
    pts = []
    for i, (s, e, k, _) in enumerate(ev):
        if e <= lo or s >= hi:
            continue
        pts.append((max(s, lo), 1, i)); pts.append((min(e, hi), -1, i))
    pts.sort()
    active, nm = set(), 0
    credit = collections.Counter()
    spans = []   # (start, length, names) of the maximal mfma-free stretches
    prev = lo
    cur = None
    for t, d, i in pts:
        if t > prev and nm == 0 and active:
            for j in active:
                credit[short_name(ev[j][3])] += (t - prev) / len(active)
            if cur is None:
                cur = [prev, 0, collections.Counter()]
            cur[1] += t - prev
            for j in active:
                cur[2][short_name(ev[j][3])] += t - prev
        if d > 0:
            active.add(i); nm += ev[i][2] == 0
            if ev[i][2] == 0 and cur is not None:
                spans.append(cur); cur = None
        else:
            active.discard(i); nm -= ev[i][2] == 0
        prev = t
    if cur is not None:
        spans.append(cur)
    print(f'time with no matrix-core kernel running, by kernel (per step, {steps} steps):')
    for k, v in credit.most_common(12):
        print(f'  {k[:60]:60s} {v / steps / 1e6:8.3f} ms')
    step_len = (hi - lo) / steps
    print('longest matrix-core-free stretches (position in its step, length, kernels):')
    for s0, ln, names in sorted(spans, key=lambda x: -x[1])[:24]:
        pos = ((s0 - lo) % step_len) / 1e6
        print(f'  @{pos:7.3f} ms  {ln / 1e3:7.1f} us  ' + ', '.join(f'{k[:34]} {v / 1e3:.0f}' for k, v in names.most_common(4)))


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
    ev = []
    for r in rows:
        name = r['Kernel_Name']
        kind = 0 if any(m in name for m in MFMA) else 1
        ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), kind, name))
    ev.sort()
    # window = whole train steps: from the first to the last loss kernel of the second half of the run (one per step)
    marks = [e[0] for e in ev if 'l2_loss_k' in e[3]]
    marks = marks[len(marks) // 2:]
    if len(marks) >= 2:
        lo, hi = marks[0], marks[-1]
        print(f'{len(marks) - 1} steps, {(hi - lo) / (len(marks) - 1) / 1e6:.3f} ms per step under the profiler')
    else:
        t0, t1 = ev[0][0], max(e[1] for e in ev)
        lo, hi = t0 + (t1 - t0) * skip, t1
    pts = []
    for s, e, k, _ in ev:
        if e <= lo or s >= hi:
            continue
        s = max(s, lo); e = min(e, hi)
        pts.append((s, 1, k)); pts.append((e, -1, k))
    pts.sort()
    cnt = [0, 0]
    acc = collections.Counter()
    prev = lo
    for t, d, k in pts:
        state = ('mfma' if cnt[0] else '') + ('+' if cnt[0] and cnt[1] else '') + ('hbm' if cnt[1] else '') or 'idle'
        acc[state] += t - prev
        prev = t
        cnt[k] += d
    acc[('mfma' if cnt[0] else '') + ('+' if cnt[0] and cnt[1] else '') + ('hbm' if cnt[1] else '') or 'idle'] += hi - prev
    tot = sum(acc.values())
    for k, v in sorted(acc.items(), key=lambda kv: -kv[1]):
        print(f'{k:10s} {v / 1e6:9.3f} ms  {100 * v / tot:5.1f} %')
    hbm_only_report(ev, lo, hi, len(marks) - 1 if len(marks) >= 2 else 1)
    dur = collections.Counter(); n = collections.Counter()
    for s, e, k, name in ev:
        if e > lo and s < hi:
            short = name.replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0]
            dur[short] += e - s; n[short] += 1
    print('sum of kernel durations / wall:', round(sum(dur.values()) / tot, 3))
    for k, v in dur.most_common(14):
        print(f'  {k[:60]:60s} {n[k]:5d} {v / 1e6:8.3f} ms  avg {v / n[k] / 1e3:7.1f} us')


if __name__ == '__main__':
    main()
