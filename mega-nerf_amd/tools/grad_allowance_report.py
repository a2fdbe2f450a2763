#!/usr/bin/env python3
"""Which tensors of a training fixture sit outside `2e-4 + 2 x (the reference's own fp32 error)` against the reference's fp64 gradients
(tests/test_gpu_parity.py check_gradients_against_reference; the counts are pinned in tests/golden/gradient_allowance.json)?  One line per
tensor.  A flipped ReLU unit in layer L of a cell shows as that cell's layers <= L all off by the same few 1e-3: two such events account for
the 15 tensors of render_joint_sh2_train.   python mega-nerf_amd/tools/grad_allowance_report.py <fixture>   (GPU box, repo root)"""
import sys, os
sys.path.insert(0, 'tests'); sys.path.insert(0, 'tests/golden'); sys.path.insert(0, '.'); sys.path.insert(0, 'mega-nerf_amd')
import numpy as np, torch
from argparse import Namespace
import common
from test_gpu_parity import T, native_models, load
import mega_nerf.training as TRN
from mega_nerf.rendering import render_rays
name = sys.argv[1]
g = load(name)
hp, nerf, bg_nerf = native_models(name)
hp = Namespace(**vars(hp))
s = common.SCENE
rnd = {k[4:]: T(v).reshape(-1) if 'noise' in k else T(v) for k, v in g.items() if k.startswith('rnd_')}
idx = T(g['idx'].astype(np.int32))
flags = [bool(v) for v in g['flags']]
TRN.FORCE_GENERAL = True
res, present = render_rays(nerf, bg_nerf, T(g['rays']), idx, hp, T(s['sphere_center']), T(s['sphere_radius']), *flags, _randoms=rnd)
loss = torch.nn.functional.mse_loss(res['rgb_fine'], T(g['target']))
loss.backward()
st = int(g['gstride']) if 'gstride' in g else 37
rows = []
for tag, m in (('fg', nerf), ('bg', bg_nerf)):
    for pn, p in m.named_parameters():
        got = p.grad.detach().cpu().numpy() if p.grad is not None else np.zeros(tuple(p.shape), np.float32)
        if 'grad_%s_%s' % (tag, pn) in g: r32 = g['grad_%s_%s' % (tag, pn)]
        else: r32, got = g['gsub_%s_%s' % (tag, pn)], got.reshape(-1)[::st]
        r64 = g['g64_%s_%s' % (tag, pn)].reshape(r32.shape)
        sc = float(np.abs(r64).max())
        if sc == 0: continue
        e64 = float(np.abs(got - r64).max()) / sc; eref = float(np.abs(r32.astype(np.float64) - r64).max()) / sc; e32 = float(np.abs(got - r32).max()) / sc
        if not e64 <= 2e-4 + 2 * eref: rows.append((tag + '.' + pn, e64, eref, e32))
for r in rows: print('%-50s vs64 %.1e ref32vs64 %.1e vs32 %.1e' % r)
print(len(rows))
