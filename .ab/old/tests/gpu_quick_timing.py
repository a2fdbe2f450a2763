"""Ad-hoc timing of the fused MLP kernel and render_rays (diagnostics; bench.py is the contract)."""
import sys, time
from argparse import Namespace
from pathlib import Path
import numpy as np, torch
ROOT = Path(__file__).resolve().parent.parent
for p in (ROOT, ROOT / 'mega-nerf_amd', ROOT / 'tests', ROOT / 'tests' / 'golden'):
    sys.path.insert(0, str(p))
import common
from oracle import nerf_oracle as O
from test_gpu_parity import native_nerf, T
from mega_nerf.rendering import render_rays_async

hp = O.make_hparams(coarse_samples=64, fine_samples=128)
fcfg, bcfg = common.model_cfg(hp, 3, 256), common.model_cfg(hp, 4, 256)
fg, bg = native_nerf(fcfg, common.make_weights(fcfg, 100, 1)), native_nerf(bcfg, common.make_weights(bcfg, 100, 2))
rng = np.random.default_rng(0)
import sys as _s
tile = int(_s.argv[1]) if len(_s.argv) > 1 else 0
fg.mfma_tile = tile; bg.mfma_tile = tile
for B in (4416, 8832, 1024 * 64, 1024 * 128, 65536 * 64):
    x = T(np.concatenate([rng.uniform(-1, 1, (B, 3)), rng.standard_normal((B, 3)), rng.integers(0, 100, (B, 1))], 1).astype(np.float32))
    with torch.no_grad():
        for _ in range(3): fg(x)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5): fg(x)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
    print('mlp fg B=%d: %.3f ms  %.1f TFLOP/s' % (B, dt * 1e3, B * 1211392 / dt / 1e12), flush=True)
from mega_nerf import ray_utils
s = common.SCENE
dev = torch.device('cuda')
d = ray_utils.get_ray_directions(s['W'], s['H'], s['fx'], s['fy'], s['cx'], s['cy'], True, dev)
rays_all = ray_utils.get_rays(d, T(s['c2w']), s['near'], s['far'], s['ray_altitude_range']).view(-1, 8)
for Nr in (1024, 8192, 65536):
    sel = torch.randperm(rays_all.shape[0], device=dev)[:Nr]
    rays = rays_all[sel].contiguous(); idx = torch.randint(0, 100, (Nr,), device=dev).float()
    h = Namespace(**vars(hp))
    with torch.no_grad():
        for _ in range(2): render_rays_async(fg, bg, rays, idx, h, T(s['sphere_center']), T(s['sphere_radius']), True, False, True)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5): r, nbg, err = render_rays_async(fg, bg, rays, idx, h, T(s['sphere_center']), T(s['sphere_radius']), True, False, True)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
    print('render N=%d: %.3f ms  %.0f rays/s  n_bg=%d' % (Nr, dt * 1e3, Nr / dt, int(nbg)), flush=True)
