#!/usr/bin/env python3
"""Which kernels of the training step lose most when memory gets slower?  The benchmark step (configs[1]: 1024 rays x (64 + 128), fg + bg
8 x 256) runs with its per-kernel HIP-event spans on while a hog (mnr_calibrate_hog: `w` workgroups streaming writes + reads) runs on a side
stream: once over 1 GiB (HBM / fabric traffic) and once over 1 MiB (cache resident: the same workgroup slots taken, no traffic), so that the
loss of CUs and the loss of memory latency / bandwidth can be told apart.  One JSON line per setting.  Diagnostics (the
driver's box ran the register-chained kernels 16-19 % slower and k_wgrad2 / k_tgemm not at all)."""
import json
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent.parent
for p in (ROOT, ROOT / 'mega-nerf_amd'):
    if str(p) not in sys.path:
        sys.path.insert(0, str(p))

import bench                                                     # noqa: E402
from mega_nerf import _native as N                               # noqa: E402
from mega_nerf.opts import get_opts_base                         # noqa: E402
from mega_nerf.training import FusedTrainStep                    # noqa: E402
import synthetic_scene as S                                      # noqa: E402
from mega_nerf import ray_utils                                  # noqa: E402


def main():
    dev = torch.device('cuda:0')
    hp = get_opts_base().parse_args(['--coarse_samples', '64', '--fine_samples', '128'])
    s = S.SCENE
    sc, sr = torch.from_numpy(s['sphere_center']).to(dev), torch.from_numpy(s['sphere_radius']).to(dev)
    d = ray_utils.get_ray_directions(s['W'], s['H'], s['fx'], s['fy'], s['cx'], s['cy'], True, dev)
    all_rays = ray_utils.get_rays(d, torch.from_numpy(s['c2w']).to(dev), s['near'], s['far'], s['ray_altitude_range']).view(-1, 8)
    g = torch.Generator(device='cpu').manual_seed(42)
    sel = torch.randperm(all_rays.shape[0], generator=g)[:1024].to(dev)
    batch = (all_rays[sel].contiguous(), torch.randint(0, s['appearance_count'], (1024,), generator=g).float().to(dev), torch.rand(1024, 3, generator=g).to(dev))
    (fg, _, _), (bg, _, _) = bench.build_models(hp, dev, 1000)
    fg.train(), bg.train()
    fs = FusedTrainStep([(fg, bg)], hp, sc, sr, 1024)
    scratch = torch.empty(1 << 30, dtype=torch.uint8, device=dev)
    side = torch.cuda.Stream(dev)
    lib = N.lib()
    steps = 30

    def measure(w, nbytes):
        for _ in range(5):
            fs([batch])
        torch.cuda.synchronize()
        if w:
            # size the hog to outlast the measured steps: one pass timed alone first
            t = time.perf_counter()
            N.check(lib.mnr_calibrate_hog(scratch.data_ptr(), nbytes, w, 1, side.cuda_stream))
            side.synchronize()
            one = time.perf_counter() - t
            passes = max(2, int(steps * 0.012 / max(one, 1e-5)) + 1)
            N.check(lib.mnr_calibrate_hog(scratch.data_ptr(), nbytes, w, passes, side.cuda_stream))
        fs.profile(steps)
        t = time.perf_counter()
        for _ in range(steps):
            fs([batch])
        torch.cuda.current_stream().synchronize()
        dt = (time.perf_counter() - t) / steps * 1e3
        hog_still_running = not side.query() if w else None
        torch.cuda.synchronize()
        sp = [fs.kernel_times(i) for i in range(steps)]
        fs.profile(0)
        return dt, {k: round(sum(x[k] for x in sp) / steps, 4) for k in sp[0]}, hog_still_running

    base = None
    for w, nbytes, what in [(0, 0, 'no hog'), (8, 1 << 30, '8 wg x 1 GiB'), (8, 2 << 20, '8 wg x 2 MiB (cache resident)'),
                            (32, 1 << 30, '32 wg x 1 GiB'), (32, 2 << 20, '32 wg x 2 MiB (cache resident)'),
                            (128, 1 << 30, '128 wg x 1 GiB'), (128, 2 << 20, '128 wg x 2 MiB (cache resident)'), (0, 0, 'no hog (again)')]:
        dt, spans, running = measure(w, nbytes)
        if base is None:
            base = spans
        print(json.dumps({'hog': what, 'ms_per_step': round(dt, 4), 'hog_outlasted_the_steps': running, 'spans_ms': spans,
                          'vs_no_hog': {k: round(spans[k] / base[k], 3) for k in ('fwd_c', 'fwd_f', 'bwd', 'head_grads', 'wgrad')}}), flush=True)


if __name__ == '__main__':
    main()
