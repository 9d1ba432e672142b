#!/usr/bin/env python3
"""One train step of a rocprofv3 kernel trace as a timeline: start (us from the step's loss kernel), duration, queue, kernel; the
gaps in which NO kernel runs, and per-queue busy time.  python tools/timeline.py <kernel_trace.csv> [step index from the end = 2]"""
import csv, sys, collections


def short(name):
    return name.replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0][:58]


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    back = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r.get('Queue_Id', '?'), r['Kernel_Name']) for r in rows)
    marks = [e[0] for e in ev if 'l2_loss_k' in e[3]]
    lo, hi = marks[-back - 1], marks[-back]
    step = [e for e in ev if lo <= e[0] < hi]
    print(f'step of {(hi - lo) / 1e3:.1f} us, {len(step)} kernels')
    busy = collections.Counter()
    end = lo
    idle = 0
    for s, e, q, n in step:
        gap = s - end
        flag = f'   <-- idle {gap / 1e3:.1f}' if gap > 1500 else ''
        if gap > 0:
            idle += gap
        print(f'{(s - lo) / 1e3:9.1f} {(e - s) / 1e3:8.1f}  q{q:>3} {short(n)}{flag}')
        busy[q] += e - s
        end = max(end, e)
    print(f'no kernel running: {idle / 1e3:.1f} us of {(hi - lo) / 1e3:.1f}')
    for q, b in busy.most_common():
        print(f'queue {q}: busy {b / 1e3:.1f} us')


if __name__ == '__main__':
    main()
