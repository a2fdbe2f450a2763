#!/usr/bin/env python3
"""Host timeline of training.JointCells (bench.py joint_cells_loop with 1 / 2 / 4 cells): interval between the joint steps' enqueues, time
inside an enqueue, the order in which the cells' loops reach the rendezvous.  Measured (round 6): the host enqueues a joint step every
0.1 / 0.7 / 6.5 ms for 1 / 2 / 4 cells against 6.3 / 12.3 / 24.6 ms of GPU time per step: the loops run far ahead of the device, the job is
GPU-bound (0.98-1.00 of the bare multi-cell step).   python mega-nerf_amd/tools/probe_joint_cells.py [cells]"""
import sys, time, json, os
sys.path.insert(0, '.'); sys.path.insert(0, 'mega-nerf_amd')
import torch, bench
from mega_nerf import training as TR
log = []
orig = TR.JointCells._step_all
def patched(self):
    t0 = time.perf_counter()
    orig(self)
    log.append((t0, time.perf_counter()))
TR.JointCells._step_all = patched
sub_log = []
orig_submit = TR.JointCells.submit
def psubmit(self, index, batch):
    sub_log.append((index, time.perf_counter()))
    return orig_submit(self, index, batch)
TR.JointCells.submit = psubmit
args = bench.parse_args(['--steps', '40'])
dev = torch.device('cuda:0'); torch.cuda.set_device(dev)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4
r = bench.joint_cells_loop(args, dev, n)
iv = [b[0] - a[0] for a, b in zip(log[25:], log[26:])]
inside = [b - a for a, b in log[25:]]
print(json.dumps({'cells': n, 'result_ms': r.get('ms_per_joint_iteration'), 'bare_ms': r.get('bare_multi_cell_step_on_its_last_batches_ms'),
                  'interval_ms_mean': round(sum(iv) / len(iv) * 1e3, 3), 'step_all_ms_mean': round(sum(inside) / len(inside) * 1e3, 3),
                  'step_all_ms_max': round(max(inside) * 1e3, 3)}))
# order / gaps of the submits of one iteration
last = sub_log[-n * 3:]
print([(i, round((t - last[0][1]) * 1e3, 2)) for i, t in last])
