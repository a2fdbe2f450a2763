#!/usr/bin/env python3
"""Weight-gradient launch in isolation: round-1 kernel (atomics, one model per launch) vs the batched slab kernel
(csrc/wgrad.hip), same tapes.  Prints max relative gradient difference and ms per launch (interleaved rounds)."""
import argparse
import ctypes as C
import json
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from mega_nerf import _native as N                          # noqa: E402
from mega_nerf.models.nerf import NeRF, ShiftedSoftplus     # noqa: E402


def make(dev, xyz_dim, rows, S, seed):
    torch.manual_seed(seed)
    m = NeRF(12, 4, 8, [4], 256, 48, False, 100, 3, xyz_dim, ShiftedSoftplus()).to(dev)
    n_rays = rows // S
    xyz = torch.rand(rows, xyz_dim, device=dev) * 2 - 1
    dirs = torch.nn.functional.normalize(torch.randn(n_rays, 3, device=dev), dim=-1)
    idx = torch.randint(0, 100, (n_rays,), device=dev).float()
    out = torch.empty(rows, 4, device=dev)
    d_out = torch.randn(rows, 4, device=dev)
    cap = (rows + 63) // 64 * 64
    tape = torch.zeros(cap * m.tape_floats_per_row(), device=dev)
    gtape = torch.zeros_like(tape)
    io = m.mlp_io(xyz, xyz_dim, dirs, 3, idx, 1, S, rows, out)
    m.evaluate_train(io, tape, cap, 0)
    desc, packed = m.packed()
    pb = m.packed_bwd()
    dheads = torch.empty(cap, 4, device=dev)
    counter = torch.zeros(1, device=dev, dtype=torch.int32)
    grads = {k: torch.zeros_like(p) for k, p in m.named_parameters()}
    g = N.MlpGradIO()
    g.tape, g.gtape, g.tape_rows, g.tape_row0 = tape.data_ptr(), gtape.data_ptr(), cap, 0
    g.d_out, g.d_out_stride, g.out, g.out_stride = d_out.data_ptr(), 4, out.data_ptr(), 4
    g.dheads, g.idx, g.idx_stride, g.idx_is_float = dheads.data_ptr(), idx.data_ptr(), 1, 1
    g.rows_per_ray, g.n_rows, g.work_counter = S, rows, counter.data_ptr()
    g.grad = m.grad_struct(grads)
    N.check(N.lib().mnr_mlp_backward_data(packed.data_ptr(), pb.data_ptr(), C.byref(desc), C.byref(g), N.stream_ptr()))
    keep = (xyz, dirs, idx, out, d_out, tape, gtape, dheads, counter, packed, pb)
    return dict(m=m, desc=desc, g=g, grads=grads, cap=cap, rows=rows, tape=tape, gtape=gtape, keep=keep)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--rows', type=int, default=196608)
    ap.add_argument('--bg-rows', type=int, default=13248)
    ap.add_argument('--rounds', type=int, default=5)
    a = ap.parse_args()
    dev = torch.device('cuda')
    lib = N.lib()
    fg = make(dev, 3, a.rows, 192, 0)
    bg = make(dev, 4, a.bg_rows, 96, 1) if a.bg_rows else None
    models = [fg] + ([bg] if bg else [])
    ws = torch.empty(lib.mnr_wgrad_workspace_bytes(), dtype=torch.uint8, device=dev)

    def run_old():
        for x in models:
            N.check(lib.mnr_mlp_backward_weights(C.byref(x['desc']), C.byref(x['g']), N.stream_ptr()))

    regs = (N.WgradRegion * len(models))()
    for r, x in zip(regs, models):
        r.desc = C.pointer(x['desc'])
        r.tape, r.gtape, r.tape_rows = x['tape'].data_ptr(), x['gtape'].data_ptr(), x['cap']
        r.n_ranges = 1
        r.row0[0], r.n_rows[0] = 0, x['rows']
        r.grad = x['g'].grad

    def run_new():
        N.check(lib.mnr_mlp_backward_weights_multi(regs, len(models), ws.data_ptr(), ws.numel(), N.stream_ptr()))

    def grads_of(fn):
        for x in models:
            for v in x['grads'].values():
                v.zero_()
        fn()
        torch.cuda.synchronize()
        return [{k: v.clone() for k, v in x['grads'].items()} for x in models]

    g_old, g_new = grads_of(run_old), grads_of(run_new)
    worst = 0.0
    for go, gn in zip(g_old, g_new):
        for k in go:
            if k.startswith(('sigma', 'rgb', 'embedding_a')):
                continue
            scale = float(go[k].abs().max()) + 1e-30
            worst = max(worst, float((go[k] - gn[k]).abs().max()) / scale)
    ctr = ws[:256].view(torch.int32).cpu().tolist()
    res = {'max_rel_diff_vs_round1_kernel': worst, 'episodes': ctr[24], 'items_pulled': ctr[:24]}

    def timed(fn, reps=5):
        a_, b_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a_.record()
        for _ in range(reps):
            fn()
        b_.record()
        torch.cuda.synchronize()
        return a_.elapsed_time(b_) / reps

    told, tnew = [], []
    for _ in range(a.rounds):
        told.append(timed(run_old))
        tnew.append(timed(run_new))
    flops = 2 * sum(x['rows'] * (sum(p.numel() for n, p in x['m'].named_parameters() if n.endswith('weight') and
                                     not n.startswith(('embedding_a', 'sigma', 'rgb')))) for x in models)
    import os
    if os.environ.get('MNR_WGRAD_PROF'):
        off = 256 + 768 * 4 + 767 * 98816 * 4
        ws[off:off + 256 * 64].zero_()
        run_new()
        torch.cuda.synchronize()
        pr = ws[off:off + 256 * 64].view(torch.int64).view(256, 8).cpu().double()
        tot = pr[:, 6]
        res['prof'] = {'wg_total_cycles_mean': float(tot.mean()), 'wg_total_min': float(tot.min()), 'wg_total_max': float(tot.max()),
                       'frac_vmwait': float((pr[:, 0] / tot).mean()), 'frac_barrier': float((pr[:, 1] / tot).mean()),
                       'frac_compute': float((pr[:, 2] / tot).mean()), 'tiles_mean': float(pr[:, 3].mean()),
                       'frac_in_episodes': float((pr[:, 4] / tot).mean()), 'episodes_per_wg': float(pr[:, 5].mean()),
                       'compute_cycles_per_tile': float((pr[:, 2].sum() / pr[:, 3].sum()))}
    res.update(old_ms=min(told), new_ms=min(tnew), old_ms_all=[round(t, 4) for t in told], new_ms_all=[round(t, 4) for t in tnew],
               gflop=flops / 1e9, old_tflops=flops / min(told) / 1e9, new_tflops=flops / min(tnew) / 1e9)
    print(json.dumps(res), flush=True)


if __name__ == '__main__':
    main()
