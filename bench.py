#!/usr/bin/env python3
"""bench.py -- rays/s of the Mega-NeRF hot path (get_rays + render_rays + NeRF MLP) on MI355X.

Contract (driver): ``python bench.py --gpus N --steps K --warmup W``; for N > 1 it is launched by
``python -m torch.distributed.run --nproc-per-node N ...`` with one rank per GPU (RCCL).  Rank 0 prints ONE
JSON line.

Workload (BASELINE.json configs[1]/[2]): "configs/mega-nerf Rubble"-shaped model -- foreground + background
NeRF, 8 layers x 256 channels, 12/4 frequency bands, 48-d appearance embedding -- on synthetic
1024-ray x (64 coarse + 128 fine)-sample batches; one spatial submodule per GPU (weak scaling: every rank
owns a private submodule and its own ray batch; no collective in the data path, one RCCL all-reduce of the
packed metric vector after the timed region, replacing the reference's file-based gather runner.py:495-510).
A "step" is one pass of the hot path over one batch: ``--mode eval`` = render_rays forward with the
validation flags (runner.py:569-578); ``--mode train`` = render_rays + MSE loss + backward + 2x Adam
(runner.py:246-277).  Inputs are resident in HBM before the timed region.
"""
import argparse
import json
import os
import sys
import time
from pathlib import Path

os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')      # dmabuf IPC: required for RCCL across processes on this host driver

import torch          # noqa: E402

ROOT = Path(__file__).resolve().parent
for p in (ROOT, ROOT / 'mega-nerf_amd', ROOT / 'tests', ROOT / 'tests' / 'golden'):
    if str(p) not in sys.path:
        sys.path.insert(0, str(p))

FG_FLOP_PER_SAMPLE = 1211392      # 2 x 605 696 MAC
BG_FLOP_PER_SAMPLE = 1236992
PEAK_F32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense


def build_models(hp, dev, seed):
    import common
    from mega_nerf.models.nerf import NeRF, ShiftedSoftplus
    A = common.SCENE['appearance_count']
    out = []
    for xyz_dim, s in ((3, seed), (4, seed + 500)):
        cfg = common.model_cfg(hp, xyz_dim, 256)
        w = common.make_weights(cfg, A, s)
        m = NeRF(cfg.pos_xyz_dim, cfg.pos_dir_dim, cfg.layers, cfg.skip_layers, cfg.layer_dim, cfg.appearance_dim,
                 False, A, 3, xyz_dim, ShiftedSoftplus())
        m.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()})
        out.append((m.to(dev), cfg, w))
    return out


def cpu_baseline(hp, rays_np, idx_np, tgt_np, fw, bw, fcfg, bcfg, n_sample, mode):
    """The reference algorithm restated with the same torch CPU ops (oracle/torch_oracle.py, pinned to the golden
    vectors) timed on this box's host cores on a bounded sample of the same batch: forward render for --mode eval,
    forward + autograd backward + 2x Adam for --mode train (runner.py:246-277)."""
    import common
    from oracle import torch_oracle as TO
    s = common.SCENE
    cores = usable_cores()
    torch.set_num_threads(cores)
    fg, bg = TO.make_models(hp, fcfg, fw, bcfg, bw, s['appearance_count'])
    sc, sr = torch.from_numpy(s['sphere_center']), torch.from_numpy(s['sphere_radius'])
    if mode == 'train':
        fg.train(), bg.train()
        opts = [torch.optim.Adam(fg.parameters(), lr=5e-4), torch.optim.Adam(bg.parameters(), lr=5e-4)]
    else:
        fg.eval(), bg.eval()

    def run(n):
        rays, idx, tgt = torch.from_numpy(rays_np[:n]), torch.from_numpy(idx_np[:n]), torch.from_numpy(tgt_np[:n])
        t0 = time.perf_counter()
        if mode == 'train':
            for o in opts:
                o.zero_grad(set_to_none=True)
            res = TO.render_rays(fg, bg, rays, idx, hp, sc, sr)
            torch.nn.functional.mse_loss(res['rgb_fine'], tgt).backward()
            for o in opts:
                o.step()
        else:
            with torch.inference_mode():
                TO.render_rays(fg, bg, rays, idx, hp, sc, sr)
        return time.perf_counter() - t0

    # bounded sample: a 32-ray probe (also the warm-up) sizes the timed sample to ~5 s per repetition, at most the batch
    probe = min(run(32), run(32))
    n = int(max(32, min(n_sample, (5.0 / max(probe / 32, 1e-9)) // 32 * 32)))
    best = probe if n == 32 else min(run(n) for _ in range(2))
    return {'value': n / best, 'unit': 'rays/s', 'cores': int(cores), 'kind': 'port',
            'sample': 'torch-CPU restatement of the reference (%s), first %d rays x (64+128) samples of the same batch, '
                      '%d threads, best of 2 after a 32-ray warm-up' % ('fwd+bwd+2xAdam step' if mode == 'train' else 'render_rays fwd, eval flags', n, cores)}


def usable_cores(cap: int = 32) -> int:
    """Host threads for the CPU baseline: the affinity mask and the cgroup CPU quota (a container on a 256-thread host may
    own far fewer), capped -- beyond a few dozen threads these GEMM sizes only thrash."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()[:2]
        if quota != 'max':
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        try:
            q = int(open('/sys/fs/cgroup/cpu/cpu.cfs_quota_us').read())
            per = int(open('/sys/fs/cgroup/cpu/cpu.cfs_period_us').read())
            if q > 0:
                n = min(n, max(1, q // per))
        except Exception:
            pass
    return max(1, min(n, cap))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=50)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--mode', choices=['eval', 'train'], default='train',
                    help='train: fwd+bwd+Adam step (BASELINE metric: train rays/s); eval: render_rays forward only')
    ap.add_argument('--rays', type=int, default=1024, help='rays per batch (BASELINE: 1024)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--container', type=int, default=0, metavar='N',
                    help='eval mode only: render through a merged N-cell container (MegaNeRF router, boundary_margin 1.15) '
                         'instead of one submodule -- the "8-submodule Rubble" evaluation shape on ONE GPU')
    args = ap.parse_args()

    rank = int(os.environ.get('RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X: the hot path has no CPU fallback')
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group('nccl', device_id=dev)            # RCCL on ROCm, communicator bound to this rank's GPU
    assert args.gpus == world, '--gpus must equal WORLD_SIZE (launch with torch.distributed.run)'

    import common                      # tests/golden/common.py: the seeded scene / weight generator (no oracle code)
    from mega_nerf import ray_utils, rendering
    from mega_nerf.opts import get_opts_base
    from mega_nerf.rendering import render_rays_async

    # opts.py defaults = configs/mega-nerf (8x256, 12/4 frequency bands, 48-d appearance) at the benchmark's 64+128 samples
    hp_o = hp = get_opts_base().parse_args(['--coarse_samples', '64', '--fine_samples', '128'])
    s = common.SCENE
    (fg, fcfg, fw), (bg, bcfg, bw) = build_models(hp_o, dev, 1000 * (rank + 1))   # one submodule per rank
    if args.container:
        assert args.mode == 'eval', '--container is an evaluation shape (routed containers are inference-only)'
        from mega_nerf.models.mega_nerf import MegaNeRF
        n = args.container
        g0 = max(1, int(round(n ** 0.5)) if int(round(n ** 0.5)) ** 2 == n else 2)
        g1 = n // g0
        assert g0 * g1 == n, '--container must factor into a grid'
        cent = torch.stack([torch.zeros(n), torch.linspace(-.45, .45, g0).repeat_interleave(g1),
                            torch.linspace(-.45, .45, g1).repeat(g0)], 1)
        cells = [build_models(hp_o, dev, 1000 * (rank + 1) + 7 * j) for j in range(n)]
        fg = MegaNeRF([c[0][0] for c in cells], cent, hp.boundary_margin, False, False).to(dev)
        bg = MegaNeRF([c[1][0] for c in cells], cent, hp.boundary_margin, True, False).to(dev)
        hp.container_path = 'bench'              # background points carry their world position for the router (quirk Q15)
        args.no_cpu_baseline = True
    sc, sr = torch.from_numpy(s['sphere_center']).to(dev), torch.from_numpy(s['sphere_radius']).to(dev)

    # synthetic batch: rays of the 400x400 camera, seeded permutation, ~13 % bg rays
    d = ray_utils.get_ray_directions(s['W'], s['H'], s['fx'], s['fy'], s['cx'], s['cy'], True, dev)
    all_rays = ray_utils.get_rays(d, torch.from_numpy(s['c2w']).to(dev), s['near'], s['far'],
                                  s['ray_altitude_range']).view(-1, 8)
    g = torch.Generator(device='cpu').manual_seed(42 + rank)
    sel = torch.randperm(all_rays.shape[0], generator=g)[:args.rays].to(dev)
    rays = all_rays[sel].contiguous()
    idx = torch.randint(0, s['appearance_count'], (args.rays,), generator=g).float().to(dev)
    target = torch.rand(args.rays, 3, generator=g).to(dev)

    if args.mode == 'train':
        from mega_nerf.training import TrainStep
        fg.train(), bg.train()
        stepper = TrainStep(fg, bg, hp, sc, sr)
        step = lambda: stepper(rays, idx, target)                       # noqa: E731
    else:
        fg.eval(), bg.eval()

        def step():
            with torch.no_grad():
                return render_rays_async(fg, bg, rays, idx, hp, sc, sr, True, False, True)

    for _ in range(args.warmup):
        step()
    # kernel-level timing of the dominant launch (fg fine MLP: rays x 128 rows) with HIP events recorded on the
    # launch stream inside the timed region
    rendering.KERNEL_EVENTS = ev = []
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    dt = time.perf_counter() - t0
    rendering.KERNEL_EVENTS = None
    tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
    n_bg = int(out[1]) if (args.mode == 'eval' and out[1] is not None) else -1
    if dist is not None:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax)

    # eval metric all-reduce (packed [sum_psnr, count]) -- the only collective of the path
    with torch.no_grad():
        res = render_rays_async(fg.eval(), bg.eval(), rays, idx, hp, sc, sr, True, False, True)[0]
        mse = torch.mean((res['rgb_fine'] - target) ** 2)
        packed = torch.stack([-10 * torch.log10(mse), torch.ones((), device=dev)]).double()
    if dist is not None:
        dist.all_reduce(packed)
    psnr = float(packed[0] / packed[1])

    # a second, untimed pass of the other mode so that one line carries both halves of the BASELINE metric
    other = None
    if rank == 0 or dist is not None:
        if args.mode == 'train':
            fg.eval(), bg.eval()
            with torch.no_grad():
                for _ in range(3):
                    render_rays_async(fg, bg, rays, idx, hp, sc, sr, True, False, True)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for _ in range(args.steps):
                    render_rays_async(fg, bg, rays, idx, hp, sc, sr, True, False, True)
                torch.cuda.synchronize()
                other = ('eval_rays_per_sec_per_gpu', args.rays * args.steps / (time.perf_counter() - t1))

    if rank == 0:
        total_rays = args.rays * args.steps * world
        traffic_file = ROOT / 'profiles' / 'hbm_traffic.json'
        traffic_tab = json.loads(traffic_file.read_text()) if traffic_file.exists() else {}

        def roofline(tag, kernel, flops, key):
            """tag: one launch tag, or a tuple of tags = every launch of one kernel symbol in a step (then ``flops`` is the
            mean per launch), so that avg_launch_ms is the same population as the kernel's row in a rocprofv3 trace."""
            tags = tag if isinstance(tag, tuple) else (tag,)
            ms = [a.elapsed_time(b) for t, a, b in ev if t in tags]
            if not ms:
                return None
            avg = sum(ms) / len(ms) * 1e-3
            ach = flops / avg / 1e12
            return {'bound': 'mfma', 'achieved': round(ach, 2), 'peak': PEAK_F32_MFMA_TFLOPS, 'unit': 'TFLOP/s',
                    'frac': round(ach / PEAK_F32_MFMA_TFLOPS, 4), 'traffic': traffic_tab.get(key), 'kernel': kernel,
                    'avg_launch_ms': round(avg * 1e3, 4), 'algorithmic_gflop_per_launch': round(flops / 1e9, 2)}

        n_fine, n_all = args.rays * 128, args.rays * 192
        if args.container:
            ev = []          # routed launches have data-dependent row counts: no per-kernel roofline for this shape
        fwd_fine = roofline('fg_fine', 'k_mlp_fwd<fg> (fine pass, %d rows)' % n_fine, n_fine * FG_FLOP_PER_SAMPLE,
                            'k_mlp_fwd_fg_fine_bytes_per_launch')
        if args.mode == 'train':
            # dominant kernel of a training step: the weight-gradient GEMMs of all fg layers in one launch
            # (algorithmic FLOPs = 2 * rows * sum_l M_l*N_l over the MFMA layers = per-sample forward MACs * 2
            #  minus the two VALU heads)
            wgrad_flops = n_all * (FG_FLOP_PER_SAMPLE - 2 * (256 + 3 * 128))
            roof = roofline('fg_wgrad', 'k_wgrad<true> (fg, %d rows x 13 layer jobs; the only launch of this symbol per step)' % n_all,
                            wgrad_flops, 'k_wgrad_fg_bytes_per_launch')
            extra_roof = {'k_mlp_fwd_train_fg_fine': fwd_fine,
                          'k_mlp_bwd_fg_fine': roofline('fg_bwd_fine', 'k_mlp_bwd<fg> (fine rows)',
                                                        n_fine * (FG_FLOP_PER_SAMPLE - 2 * 80 * 256 - 2 * (27 * 128)),
                                                        'k_mlp_bwd_fg_fine_bytes_per_launch')}
        else:
            # every launch of the fg forward symbol in a step (coarse 64 + fine 128 samples per ray): the population a
            # rocprofv3 kernel trace averages over; the fine pass alone is reported beside it
            n_coarse = args.rays * 64
            roof = roofline(('fg_coarse', 'fg_fine'), 'k_mlp_fwd<fg, false> (coarse %d + fine %d rows: both launches per step)' % (n_coarse, n_fine),
                            (n_coarse + n_fine) / 2 * FG_FLOP_PER_SAMPLE, 'k_mlp_fwd_fg_all_bytes_per_launch')
            extra_roof = {'k_mlp_fwd_fg_fine_only': fwd_fine}
        cpu = None
        if not args.no_cpu_baseline and world == 1:          # rank 0 at N = 1 only
            cpu = cpu_baseline(hp_o, rays.cpu().numpy(), idx.cpu().numpy(), target.cpu().numpy(), fw, bw, fcfg, bcfg,
                               min(1024, args.rays), args.mode)
        line = {
            'metric': 'train rays/sec (fwd+bwd+2xAdam step)' if args.mode == 'train' else 'eval rays/sec (render_rays fwd)',
            'value': total_rays / dt, 'unit': 'rays/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': dt / args.steps * 1e3, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': 'configs/mega-nerf Rubble-shaped fg+bg NeRF (8x256, 12/4 freqs, 48-d appearance), '
                                   '%d rays x (64+128) samples per step, %s' % (
                                       args.rays, 'one submodule per GPU' if not args.container else
                                       'merged %d-cell container (MegaNeRF router, margin 1.15)' % args.container),
                       'mode': args.mode, 'rays_per_batch': args.rays, 'bg_rays_in_batch': n_bg,
                       'parallelism': 'submodule-per-gpu x%d' % world if not args.container else
                       '%d-cell container routed on one GPU' % args.container},
            'eval_psnr_vs_random_target_db': round(psnr, 4),
            'roofline': roof, 'cpu_baseline': cpu,
        }
        if extra_roof and any(v is not None for v in extra_roof.values()):
            line['roofline_other_kernels'] = extra_roof
        if other:
            line[other[0]] = other[1]
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
