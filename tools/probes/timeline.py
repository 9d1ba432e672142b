import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Queue_Id'], r['Kernel_Name'], r['VGPR_Count']) for r in rows)
# last step: find the last nchw_to_nhwc group start
idx = [i for i, e in enumerate(ev) if 'l2_loss_k' in e[3]]
start = idx[-2] if len(idx) > 1 else 0
t0 = ev[start][0]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 200
for s, e, q, name, v in ev[start:start + n]:
    short = name.replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0][:44]
    print(f'{(s - t0) / 1e3:9.1f} {(e - t0) / 1e3:9.1f} {(e - s) / 1e3:7.1f} q{q} v{v} {short}')
