#!/usr/bin/env python3
"""Pins bench.py's CPU baseline: times oracle/torch_oracle.py (the torch-CPU restatement that bench.py runs on the GPU box, where
/root/reference does not exist: ``cpu_baseline.kind = "port"``) against the REAL reference (mega_nerf.rendering.render_rays of
/root/reference, torch CPU fp32) on the benchmark's workload -- 1024 rays x (64 + 128) samples, fg + bg 8x256 models -- eval
(render_rays under inference_mode) and train (render_rays + mse_loss + backward + Adam on both models, runner.py:244-277).

BUILD CONTAINER ONLY (imports /root/reference).  Usage:  python mega-nerf_amd/tools/cpu_port_vs_reference.py > profiles/rNN_cpu_port_vs_reference.json
"""
import json
import os
import sys
import time
from argparse import Namespace
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[2]
sys.dont_write_bytecode = True
sys.path.insert(0, str(ROOT))
sys.path.insert(0, '/root/reference')                          # FIRST: its package is also called mega_nerf
sys.path.append(str(ROOT / 'mega-nerf_amd'))                   # synthetic_scene only (the product's mega_nerf stays shadowed)

from mega_nerf import rendering as REF                         # noqa: E402  (the reference)
from mega_nerf import ray_utils as RU                          # noqa: E402
from mega_nerf.models import model_utils as MU                 # noqa: E402

import synthetic_scene as S                                    # noqa: E402
from oracle import torch_oracle as TO                          # noqa: E402
from oracle.nerf_oracle import make_hparams                    # noqa: E402


def best_of(fn, reps=3, warm=1):
    for _ in range(warm):
        fn()
    ts = []
    for _ in range(reps):
        t = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t)
    return min(ts)


def main():
    threads = int(os.environ.get('THREADS', os.cpu_count() or 8))
    torch.set_num_threads(threads)
    s = S.SCENE
    hp = Namespace(**vars(make_hparams(coarse_samples=64, fine_samples=128)))
    A = s['appearance_count']
    fcfg, bcfg = S.model_cfg(hp, 3, 256), S.model_cfg(hp, 4, 256)
    fw, bw = S.make_weights(fcfg, A, 1000), S.make_weights(bcfg, A, 1500)
    d = RU.get_ray_directions(s['W'], s['H'], s['fx'], s['fy'], s['cx'], s['cy'], True, torch.device('cpu'))
    rays_all = RU.get_rays(d, torch.from_numpy(s['c2w']), s['near'], s['far'], s['ray_altitude_range']).view(-1, 8).numpy()
    N = int(os.environ.get('RAYS', 1024))
    rays_np, idx_np = S.pick_rays(rays_all, N, 7)
    rays = torch.from_numpy(rays_np)
    tgt = torch.rand(N, 3, generator=torch.Generator().manual_seed(1))
    sc, sr = torch.from_numpy(s['sphere_center']), torch.from_numpy(s['sphere_radius'])

    def ref_models():
        out = []
        for cfg, w in ((fcfg, fw), (bcfg, bw)):
            m = MU._get_single_nerf_inner(hp, A, cfg.layer_dim, cfg.xyz_dim)
            m.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()})
            out.append(m)
        return out

    res = {'threads': threads, 'rays': N, 'samples': '64+128', 'torch': torch.__version__}
    # ---- eval ----
    rf, rb = ref_models()
    rf.eval(), rb.eval()
    pf, pb = TO.make_models(hp, fcfg, fw, bcfg, bw, A)
    pf.eval(), pb.eval()
    idx_f = torch.from_numpy(idx_np.astype(np.float32))

    def ref_eval():
        with torch.inference_mode():
            REF.render_rays(rf, rb, rays, idx_f, hp, sc, sr, True, False, True)

    def port_eval():
        with torch.no_grad():
            TO.render_rays(pf, pb, rays, idx_f, hp, sc, sr)
    t_ref, t_port = best_of(ref_eval), best_of(port_eval)
    res['eval'] = {'reference_rays_per_s': N / t_ref, 'port_rays_per_s': N / t_port, 'port_over_reference': t_ref / t_port}
    # ---- train ----
    rf, rb = ref_models()
    rf.train(), rb.train()
    pf, pb = TO.make_models(hp, fcfg, fw, bcfg, bw, A)
    pf.train(), pb.train()
    idx_i = torch.from_numpy(idx_np.astype(np.int32))
    ro = [torch.optim.Adam(m.parameters(), lr=5e-4) for m in (rf, rb)]
    po = [torch.optim.Adam(m.parameters(), lr=5e-4) for m in (pf, pb)]

    def ref_train():
        for o in ro:
            o.zero_grad(set_to_none=True)
        out, _ = REF.render_rays(rf, rb, rays, idx_i, hp, sc, sr, False, True, False)
        torch.nn.functional.mse_loss(out['rgb_fine'], tgt).backward()
        for o in ro:
            o.step()

    def port_train():
        for o in po:
            o.zero_grad(set_to_none=True)
        out = TO.render_rays(pf, pb, rays, idx_i, hp, sc, sr)
        torch.nn.functional.mse_loss(out['rgb_fine'], tgt).backward()
        for o in po:
            o.step()
    t_ref, t_port = best_of(ref_train), best_of(port_train)
    res['train'] = {'reference_rays_per_s': N / t_ref, 'port_rays_per_s': N / t_port, 'port_over_reference': t_ref / t_port}
    print(json.dumps(res, indent=1))


if __name__ == '__main__':
    main()
