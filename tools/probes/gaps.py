"""Idle gaps of one train step in a rocprofv3 kernel trace: python tools/probes/gaps.py kernel_trace.csv [min_gap_us]"""
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Queue_Id'], r['Kernel_Name']) for r in rows)
idx = [i for i, e in enumerate(ev) if 'l2_loss_k' in e[3]]
a, b = idx[-2], idx[-1]
step = ev[a:b]
t0, t1 = step[0][0], ev[b][0]
short = lambda n: re.sub(r'\(anonymous namespace\)::|void |\(.*|v4::|v8::', '', n)[:40]
# union of busy intervals
cur_end = t0; idle = 0; gaps = []
last = None
for s, e, q, n in step:
    if s > cur_end:
        gaps.append((s - cur_end, cur_end - t0, short(last[3]) if last else '', short(n)))
        idle += s - cur_end
    if e > cur_end:
        cur_end = e; last = (s, e, q, n)
print(f'step {(t1 - t0) / 1e3:.1f} us, {len(step)} kernels, idle (no kernel on any queue) {idle / 1e3:.1f} us')
mg = float(sys.argv[2]) if len(sys.argv) > 2 else 3.0
for g, at, prev, nxt in sorted(gaps, reverse=True)[:25]:
    if g / 1e3 >= mg: print(f'  gap {g / 1e3:6.1f} us at {at / 1e3:8.1f}: after {prev} -> before {nxt}')
