#!/usr/bin/env python3
"""Why does the config-2 leg read 2-3 % lower inside the default bench.py run (after two model-F legs in the same process) than as
a process of its own?  Runs the config-2 leg  (a) first in the process, (b) right after a model-F leg, (c) after a model-F leg and
an idle pause, (d) again -- each with bench.py's own _run_leg.   python tools/probes/leg_order.py [pause seconds = 5]"""
import copy, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.argv = sys.argv[:1] + ['--no-cpu-baseline', '--extra', '0']
import torch
import bench

pause = float(os.environ.get('KP_PAUSE', '5'))
args = bench.parse()
dev = torch.device('cuda', 0)
torch.cuda.set_device(0)
c2 = copy.copy(args)
for k, v in dict(net='transporter', model_type='VGG_PONG_LAYERNECK', model_in_channels=1, model_z_channels=16, model_keypoints=4,
                 size=84, batch=256, precision='bf16', steps=100, warmup=16, profile_steps=0).items():
    setattr(c2, k, v)
f = copy.copy(args); f.steps, f.warmup, f.profile_steps = 20, 10, 0


def leg(a, tag):
    r = bench.run_leg(a, 0, 1, dev)
    print(f'{tag:42s} {r["value"]:10.1f} pairs/s  {r["ms_per_step"]:.4f} ms  median {r["ms_per_step_median_hipevent"]:.4f}  max {r["ms_per_step_max_hipevent"]:.3f}', flush=True)


leg(c2, '(a) config 2, first in the process')
leg(c2, '(a2) config 2 again')
leg(f, '    model F fp16')
leg(c2, '(b) config 2 right after model F')
leg(f, '    model F fp16')
time.sleep(pause)
leg(c2, f'(c) config 2 after model F + {pause:.0f} s idle')
leg(c2, '(d) config 2 again')
