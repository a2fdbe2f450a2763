#!/usr/bin/env python3
"""Sweep the work-item parameters of the weight-gradient launch (k_wgrad<true>, 196 608 fg rows) -- the host reads
MNR_WGRAD_ITEMS / MNR_WGRAD_FIXED / MNR_WGRAD_FILL_BPC on every call, so one process can try them all.  Prints one JSON
line per setting (HIP-event time of the whole backward = data-gradient chain + head gradients + weight gradients, and of
the weight-gradient launch alone).  Diagnostics for tuning; bench.py is the contract."""
import argparse
import ctypes as C
import itertools
import json
import os
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from mega_nerf import _native as N                          # noqa: E402
from mega_nerf.models.nerf import NeRF, ShiftedSoftplus     # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--rows', type=int, default=196608)
    ap.add_argument('--items', type=int, nargs='+', default=[768, 1024, 1536, 2048, 3072])
    ap.add_argument('--fixed', type=float, nargs='+', default=[2500.0])
    ap.add_argument('--fill_bpc', type=float, nargs='+', default=[6.0])
    ap.add_argument('--reps', type=int, default=5)
    a = ap.parse_args()
    dev = torch.device('cuda')
    torch.manual_seed(0)
    m = NeRF(12, 4, 8, [4], 256, 48, False, 100, 3, 3, ShiftedSoftplus()).to(dev)
    S = 192
    rows, n_rays = a.rows, a.rows // S
    xyz = torch.rand(rows, 3, device=dev) * 2 - 1
    dirs = torch.nn.functional.normalize(torch.randn(n_rays, 3, device=dev), dim=-1)
    idx = torch.randint(0, 100, (n_rays,), device=dev).float()
    out = torch.empty(rows, 4, device=dev)
    d_out = torch.randn(rows, 4, device=dev)
    tape = m.train_eval(xyz, 3, dirs, 3, S, idx, 1, S, rows, out, None, -1, None, 0)
    grads = {k: torch.zeros_like(p) for k, p in m.named_parameters()}
    tape.backward(d_out, 4, grads)                            # warm-up: fills the gradient tape the sweeps re-use
    desc, _ = m.packed()
    gtape = torch.empty(tape.tape.numel(), device=dev)
    counter = torch.zeros(1, device=dev, dtype=torch.int32)
    g = N.MlpGradIO()
    g.tape, g.gtape, g.tape_rows, g.tape_row0 = tape.tape.data_ptr(), gtape.data_ptr(), tape.tape_rows, 0
    g.dheads = torch.empty(rows, 4, device=dev).data_ptr()
    g.rows_per_ray, g.n_rows, g.work_counter = S, rows, counter.data_ptr()
    g.grad = m.grad_struct(grads)
    for items, fixed, bpc in itertools.product(a.items, a.fixed, a.fill_bpc):
        os.environ.update(MNR_WGRAD_ITEMS=str(items), MNR_WGRAD_FIXED=str(fixed), MNR_WGRAD_FILL_BPC=str(bpc))
        N.check(N.lib().mnr_mlp_backward_weights(C.byref(desc), C.byref(g), N.stream_ptr()))
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.reps):
            N.check(N.lib().mnr_mlp_backward_weights(C.byref(desc), C.byref(g), N.stream_ptr()))
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / a.reps
        flops = rows * (1211392 - 2 * (256 + 3 * 128))
        print(json.dumps({'items': items, 'fixed': fixed, 'fill_bpc': bpc, 'wgrad_ms': round(ms, 4), 'tflops': round(flops / ms / 1e9, 1)}),
              flush=True)


if __name__ == '__main__':
    main()
