#!/usr/bin/env python3
"""Step time of the one-call training step under the three forward schedules (one stream; background branch forked behind the sample
kernel; background branch forked behind the foreground's coarse pass), fp32 and split precision.  Each schedule runs in its own
process (the plan reads the environment when it is made)."""
import json
import os
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[2]
CHILD = r'''
import sys, time, json, os
from argparse import Namespace
sys.path.insert(0, %r); sys.path.insert(0, %r)
import numpy as np, torch
import synthetic_scene as S
from mega_nerf import ray_utils
from mega_nerf.opts import get_opts_base
from mega_nerf.models.nerf import NeRF, ShiftedSoftplus
from mega_nerf.training import FusedTrainStep
dev = torch.device('cuda'); s = S.SCENE
hp = get_opts_base().parse_args(['--coarse_samples', '64', '--fine_samples', '128'])
A = s['appearance_count']
def mk(xyz, seed):
    cfg = S.model_cfg(hp, xyz, 256); w = S.make_weights(cfg, A, seed)
    m = NeRF(cfg.pos_xyz_dim, cfg.pos_dir_dim, cfg.layers, cfg.skip_layers, cfg.layer_dim, cfg.appearance_dim, False, A, 3, xyz, ShiftedSoftplus())
    m.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()}); return m.to(dev).train()
d = ray_utils.get_ray_directions(s['W'], s['H'], s['fx'], s['fy'], s['cx'], s['cy'], True, dev)
allr = ray_utils.get_rays(d, torch.from_numpy(s['c2w']).to(dev), s['near'], s['far'], s['ray_altitude_range']).view(-1, 8)
g = torch.Generator(device='cpu').manual_seed(42)
sel = torch.randperm(allr.shape[0], generator=g)[:1024].to(dev)
batch = (allr[sel].contiguous(), torch.randint(0, A, (1024,), generator=g).float().to(dev), torch.rand(1024, 3, generator=g).to(dev))
sc, sr = torch.from_numpy(s['sphere_center']).to(dev), torch.from_numpy(s['sphere_radius']).to(dev)
out = {}
for split in (False, True):
    st = FusedTrainStep([(mk(3, 1000), mk(4, 1500))], hp, sc, sr, 1024, split_precision=split)
    for _ in range(10): st([batch])
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(100): st([batch])
    torch.cuda.synchronize(); ms = (time.perf_counter() - t) * 10
    st.profile(16)
    for _ in range(16): st([batch])
    torch.cuda.synchronize()
    sp = [st.kernel_times(i) for i in range(16)]
    out['split' if split else 'f32'] = {'ms_per_step': round(ms, 4), 'spans': {k: round(sum(x[k] for x in sp) / 16, 4) for k in sp[0]}}
    del st
print(json.dumps(out))
''' % (str(ROOT), str(ROOT / 'mega-nerf_amd'))

res = {}
for name, env in (('one_stream', {'MNR_STEP_ONE_STREAM': '1'}), ('fork_behind_samples', {'MNR_STEP_TWO_STREAMS': '1'}),
                  ('fork_behind_fg_coarse', {'MNR_STEP_TWO_STREAMS': '2'})):
    e = dict(os.environ)
    for k in ('MNR_STEP_ONE_STREAM', 'MNR_STEP_TWO_STREAMS'):
        e.pop(k, None)
    e.update(env)
    r = subprocess.run([sys.executable, '-c', CHILD], env=e, capture_output=True, text=True)
    line = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    res[name] = json.loads(line[0]) if line else {'error': r.stderr[-500:]}
    print(name, json.dumps(res[name]), flush=True)
