#!/usr/bin/env python3
"""bench.py -- rays/s of the Mega-NeRF hot path (get_rays + render_rays + NeRF MLP) on MI355X.

Contract (driver): ``python bench.py --gpus N --steps K --warmup W``; for N > 1 it is launched by
``python -m torch.distributed.run --nproc-per-node N ...`` with one rank per GPU (RCCL).  Rank 0 prints ONE
JSON line.

Workload (BASELINE.json configs[1]/[2]): "configs/mega-nerf Rubble"-shaped model -- foreground + background
NeRF, 8 layers x 256 channels, 12/4 frequency bands, 48-d appearance embedding -- on synthetic
1024-ray x (64 coarse + 128 fine)-sample batches.  A "step" is one pass of the hot path over one batch:
``--mode train`` = render_rays + MSE loss + backward + 2x Adam (runner.py:246-277); ``--mode eval`` =
render_rays forward with the validation flags (runner.py:569-578).  Inputs are resident in HBM before the
timed region.

Sharding: the path partitions by spatial submodule, no collective in the data path; the only RCCL
traffic is one all-reduce of the packed metric vector after the timed region (replacing the reference's
file-based gather, runner.py:495-510).
  * default: one private submodule + ray batch per rank (``"scaling": "weak"``, parscripts/run_8.txt);
  * ``--submodules S``: a fixed set of S submodules (Rubble: 8) dealt to the ranks round-robin, every rank steps
    through its cells one after the other (``"scaling": "strong"``; --gpus 1 = one GPU trains all S).

The line also carries: ``roofline`` (dominant kernel: live HIP-event duration of its launches, algorithmic FLOPs,
MFMA-busy / HBM traffic from the committed PMC summary ``profiles/r06_pmc_summary.json``, three timed regions, the box's calibration), ``cpu_baseline`` (the
torch-CPU restatement of the reference on this box's host cores, bounded sample), the north-star PSNR check
(``psnr``: a student model trained for a few steps here and by the CPU restatement on identical batches and
random numbers, both evaluated against a fixed teacher field), and (N = 1) short extra measurements: 65 536-ray
evaluation batches and the reference-default 256+512 samples per ray (opts.py:32-35,75).
"""
import argparse
import json
import os
import sys
import time
from pathlib import Path

os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')      # dmabuf IPC: required for RCCL across processes on this host driver
# Kernel arguments in DEVICE memory (this ROCm's default; read by the HIP runtime when it initialises): with HIP_FORCE_DEV_KERNARG=0 every
# wavefront's argument reads go to host memory -- measured here: the ray-stage kernels +4 ... +27 %, the step +0.8 % on a healthy box
# (profiles/r06_kernarg_probe.txt), more where the host link is slow.  Pinned so that a stray setting of the launching shell cannot decide
# the measurement; MNR_BENCH_KERNARG=0 measures the other placement.
os.environ['HIP_FORCE_DEV_KERNARG'] = os.environ.get('MNR_BENCH_KERNARG', '1')

import torch          # noqa: E402

ROOT = Path(__file__).resolve().parent
for p in (ROOT, ROOT / 'mega-nerf_amd'):
    if str(p) not in sys.path:
        sys.path.insert(0, str(p))

FG_FLOP_PER_SAMPLE = 1211392      # 2 x 605 696 MAC
BG_FLOP_PER_SAMPLE = 1236992
HEAD_FLOP_PER_SAMPLE = 2 * (256 + 3 * 128)          # sigma / rgb heads: VALU, not part of the MFMA kernels' work
PEAK_F32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 / 16x16x4_f32, dense
PMC_FILE = ROOT / 'profiles' / 'r06_pmc_summary.json'


def build_models(hp, dev, seed, layer_dim=256):
    """(model, cfg, numpy state_dict) for the foreground and the background NeRF with seeded weights."""
    import synthetic_scene as S
    from mega_nerf.models.nerf import NeRF, ShiftedSoftplus
    A = S.SCENE['appearance_count']
    out = []
    for xyz_dim, s in ((3, seed), (4, seed + 500)):
        cfg = S.model_cfg(hp, xyz_dim, layer_dim if xyz_dim == 3 else hp.bg_layer_dim)     # opts.py:48-49: --layer_dim is the foreground's
        w = S.make_weights(cfg, A, s)
        m = NeRF(cfg.pos_xyz_dim, cfg.pos_dir_dim, cfg.layers, cfg.skip_layers, cfg.layer_dim, cfg.appearance_dim,
                 False, A, cfg.rgb_dim, xyz_dim, ShiftedSoftplus())
        m.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()})
        out.append((m.to(dev), cfg, w))
    return out


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


# ---- north-star PSNR check: identical student training here and in the CPU restatement -------------------------------

PSNR_STEPS, PSNR_BATCH, PSNR_TEST_RAYS = 24, 192, 512


def _port_over_reference():
    """Speed of oracle/torch_oracle.py relative to the real reference, as measured by tools/cpu_port_vs_reference.py."""
    try:
        d = json.loads((ROOT / 'profiles' / 'r06_cpu_port_vs_reference.json').read_text())
        return {m: '%.2fx / %.2fx' % (d['threads_8'][m]['port_over_reference'], d['threads_16'][m]['port_over_reference']) for m in ('train', 'eval')}
    except Exception:
        return {}


PORT_OVER_REFERENCE = _port_over_reference()


def psnr_problem(hp, all_rays_np):
    """Seeded teacher / student weights, training batches with their random numbers, held-out test rays (all numpy)."""
    import numpy as np
    import synthetic_scene as S
    rng = np.random.default_rng(20260925)
    A = S.SCENE['appearance_count']
    fcfg, bcfg = S.model_cfg(hp, 3, 256), S.model_cfg(hp, 4, 256)
    teacher = (S.make_weights(fcfg, A, 777), S.make_weights(bcfg, A, 778))
    for w in teacher:                  # a random-init colour head renders almost uniform grey: give the teacher field contrast
        w['rgb.weight'] = (w['rgb.weight'] * np.float32(16)).astype(np.float32)
    student = (S.make_weights(fcfg, A, 901, sharpen=False), S.make_weights(bcfg, A, 902, sharpen=False))
    perm = rng.permutation(all_rays_np.shape[0])
    test = np.ascontiguousarray(all_rays_np[perm[:PSNR_TEST_RAYS]])
    pool = perm[PSNR_TEST_RAYS:]
    Nc, Nf = hp.coarse_samples, hp.fine_samples
    batches = []
    for s in range(PSNR_STEPS):
        sel = pool[s * PSNR_BATCH:(s + 1) * PSNR_BATCH]
        B = PSNR_BATCH
        rnd = {'fg_perturb': rng.random((B, Nc), dtype=np.float32), 'fg_noise_coarse': rng.random(B * Nc, dtype=np.float32),
               'fg_u': rng.random((B, Nf), dtype=np.float32), 'fg_noise_fine': rng.random(B * Nf, dtype=np.float32),
               'bg_perturb': rng.random((B, Nc // 2), dtype=np.float32), 'bg_noise_coarse': rng.random(B * (Nc // 2), dtype=np.float32),
               'bg_u': rng.random((B, Nf // 2), dtype=np.float32), 'bg_noise_fine': rng.random(B * (Nf // 2), dtype=np.float32)}
        batches.append((np.ascontiguousarray(all_rays_np[sel]), rnd))
    return dict(fcfg=fcfg, bcfg=bcfg, teacher=teacher, student=student, test=test, batches=batches,
                idx_train=np.zeros(PSNR_BATCH, np.float32), idx_test=np.zeros(PSNR_TEST_RAYS, np.float32))


def _psnr(a: torch.Tensor, b: torch.Tensor) -> float:
    return float(-10.0 * torch.log10(torch.mean((a.double() - b.double()) ** 2)))      # metrics.py:8-10


def psnr_gpu(hp, prob, dev, split_step=False):
    """Teacher targets + student training on the MI355X path; returns (psnr_db, targets_train, target_test) as CPU tensors.
    ``split_step``: the student trains through the opt-in split-precision fused step (same injected random numbers, Adam 5e-4)."""
    import synthetic_scene as S
    from mega_nerf.models.nerf import NeRF, ShiftedSoftplus
    from mega_nerf.rendering import render_rays_async
    from mega_nerf.training import render_rays_train
    s = S.SCENE
    A = s['appearance_count']
    sc, sr = torch.from_numpy(s['sphere_center']).to(dev), torch.from_numpy(s['sphere_radius']).to(dev)

    def mk(cfg, w):
        m = NeRF(cfg.pos_xyz_dim, cfg.pos_dir_dim, cfg.layers, cfg.skip_layers, cfg.layer_dim, cfg.appearance_dim, False, A, 3,
                 cfg.xyz_dim, ShiftedSoftplus())
        m.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()})
        return m.to(dev)

    tf, tb = mk(prob['fcfg'], prob['teacher'][0]).eval(), mk(prob['bcfg'], prob['teacher'][1]).eval()
    idx_tr, idx_te = torch.from_numpy(prob['idx_train']).to(dev), torch.from_numpy(prob['idx_test']).to(dev)
    with torch.no_grad():
        tgt_train = [render_rays_async(tf, tb, torch.from_numpy(r).to(dev), idx_tr, hp, sc, sr, False, False, False)[0]['rgb_fine']
                     for r, _ in prob['batches']]
        tgt_test = render_rays_async(tf, tb, torch.from_numpy(prob['test']).to(dev), idx_te, hp, sc, sr, False, False, False)[0]['rgb_fine']
    sf, sb = mk(prob['fcfg'], prob['student'][0]).train(), mk(prob['bcfg'], prob['student'][1]).train()
    if split_step:
        from mega_nerf.training import FusedTrainStep
        n = prob['batches'][0][0].shape[0]
        fs = FusedTrainStep([(sf, sb)], hp, sc, sr, n, lr=5e-4, lr_decay_factor=1.0, split_precision=True)
        for (r, rnd), tgt in zip(prob['batches'], tgt_train):
            fs([(torch.from_numpy(r).to(dev), idx_tr, tgt)],
               _randoms=[{k: torch.from_numpy(v).to(dev).reshape(-1) if 'noise' in k else torch.from_numpy(v).to(dev) for k, v in rnd.items()}])
        torch.cuda.synchronize()
        del fs
        sf.eval(), sb.eval()
        with torch.no_grad():
            out = render_rays_async(sf, sb, torch.from_numpy(prob['test']).to(dev), idx_te, hp, sc, sr, False, False, False)[0]['rgb_fine']
        return _psnr(out, tgt_test), [t.cpu() for t in tgt_train], tgt_test.cpu()
    opts = [torch.optim.Adam(sf.parameters(), lr=5e-4), torch.optim.Adam(sb.parameters(), lr=5e-4)]
    for (r, rnd), tgt in zip(prob['batches'], tgt_train):
        for o in opts:
            o.zero_grad(set_to_none=True)
        res, _, _ = render_rays_train(sf, sb, torch.from_numpy(r).to(dev), idx_tr, hp, sc, sr, False, True, False,
                                      {k: torch.from_numpy(v).to(dev) for k, v in rnd.items()})
        torch.nn.functional.mse_loss(res['rgb_fine'], tgt).backward()
        for o in opts:
            o.step()
    sf.eval(), sb.eval()
    with torch.no_grad():
        out = render_rays_async(sf, sb, torch.from_numpy(prob['test']).to(dev), idx_te, hp, sc, sr, False, False, False)[0]['rgb_fine']
    return _psnr(out, tgt_test), [t.cpu() for t in tgt_train], tgt_test.cpu()


def psnr_cpu(hp, prob, tgt_train, tgt_test):
    """The same student training in the torch-CPU restatement of the reference (oracle/torch_oracle.py)."""
    import synthetic_scene as S
    from oracle import torch_oracle as TO
    s = S.SCENE
    sc, sr = torch.from_numpy(s['sphere_center']), torch.from_numpy(s['sphere_radius'])
    sf, sb = TO.make_models(hp, prob['fcfg'], prob['student'][0], prob['bcfg'], prob['student'][1], s['appearance_count'])
    sf.train(), sb.train()
    opts = [torch.optim.Adam(sf.parameters(), lr=5e-4), torch.optim.Adam(sb.parameters(), lr=5e-4)]
    idx_tr, idx_te = torch.from_numpy(prob['idx_train']), torch.from_numpy(prob['idx_test'])
    for (r, rnd), tgt in zip(prob['batches'], tgt_train):
        for o in opts:
            o.zero_grad(set_to_none=True)
        res = TO.render_rays(sf, sb, torch.from_numpy(r), idx_tr, hp, sc, sr, {k: torch.from_numpy(v) for k, v in rnd.items()})
        torch.nn.functional.mse_loss(res['rgb_fine'], tgt).backward()
        for o in opts:
            o.step()
    sf.eval(), sb.eval()
    with torch.inference_mode():
        out = TO.render_rays(sf, sb, torch.from_numpy(prob['test']), idx_te, hp, sc, sr)['rgb_fine']
    return _psnr(out, tgt_test)


# ---- CPU baseline ----------------------------------------------------------------------------------------------------

def cpu_baseline(hp, rays_np, idx_np, tgt_np, fw, bw, fcfg, bcfg, n_sample, mode, psnr_job=None):
    """The reference algorithm restated with the same torch CPU ops (oracle/torch_oracle.py, pinned to the golden
    vectors) timed on this box's host cores on a bounded sample of the same batch: forward render for --mode eval,
    forward + autograd backward + 2x Adam for --mode train (runner.py:246-277).  Also runs the CPU half of the PSNR
    check (``psnr_job``) -- the only other place bench.py touches oracle/."""
    import synthetic_scene as S
    from oracle import torch_oracle as TO
    s = S.SCENE
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
    out = {'value': n / best, 'unit': 'rays/s', 'cores': int(cores), 'kind': 'port',
           'sample': 'torch-CPU port of the reference %s, first %d rays of the batch, %d threads' % ('train step' if mode == 'train' else 'eval render', n, cores),
           'sample_detail': 'torch-CPU restatement of the reference (%s), first %d rays x (%d+%d) samples of the same batch, '
                     '%d threads, best of 2 after a 32-ray warm-up; the port runs at %s the speed of the REAL reference on this workload '
                     '(8 / 16 threads, build container: profiles/r06_cpu_port_vs_reference.json)' % (
                         'fwd+bwd+2xAdam step' if mode == 'train' else 'render_rays fwd, eval flags', n, hp.coarse_samples, hp.fine_samples, cores,
                         PORT_OVER_REFERENCE.get(mode, '?'))}
    if psnr_job is not None:
        prob, tgt_train, tgt_test = psnr_job
        t0 = time.perf_counter()
        out['psnr_db'] = psnr_cpu(hp, prob, tgt_train, tgt_test)
        out['psnr_seconds'] = round(time.perf_counter() - t0, 1)
    return out


def cpu_baseline_container(hp, rays_np, idx_np, cells):
    """Routed evaluation through a merged container on the host: the numpy restatement of the reference (oracle/nerf_oracle.py,
    pinned to the container goldens; the torch restatement has no MegaNeRF router).  BLAS threads as numpy finds them; a
    bounded sample of the same batch."""
    import numpy as np
    import synthetic_scene as S
    from oracle import nerf_oracle as O
    s = S.SCENE
    cores = usable_cores()
    fg = O.Model(cells['fcfg'], subs=cells['fg'], centroids=cells['cent'], boundary_margin=float(hp.boundary_margin), xyz_real=False, cluster_2d=False)
    bg = O.Model(cells['bcfg'], subs=cells['bg'], centroids=cells['cent'], boundary_margin=float(hp.boundary_margin), xyz_real=True, cluster_2d=False)
    ohp = O.make_hparams(**{k: getattr(hp, k) for k in vars(O.make_hparams()) if hasattr(hp, k)})
    ohp.perturb, ohp.container_path = 0.0, 'bench'

    def run(n):
        t = time.perf_counter()
        O.render_rays(fg, bg, rays_np[:n], idx_np[:n], ohp, s['sphere_center'], s['sphere_radius'], get_depth=True,
                      get_depth_variance=False, get_bg_fg_rgb=True)
        return time.perf_counter() - t
    run(8)                                   # warm-up (first-call overheads of numpy / BLAS)
    t_probe = run(16)
    n = int(max(16, min(rays_np.shape[0], 16 * 10.0 / max(t_probe, 1e-3))))          # ~10 s of host work
    dt = run(n)
    return {'value': n / dt, 'unit': 'rays/s', 'cores': cores, 'kind': 'port',
            'sample': 'numpy port of the reference routed-container render, first %d rays of the batch' % n,
            'sample_detail': 'numpy restatement of the reference (render_rays through the routed %d-cell container, eval flags), first %d rays '
                      'x (%d+%d) samples of the same batch, one run after a 16-ray probe' % (len(cells['fg']), n, hp.coarse_samples, hp.fine_samples)}



# ---- self-diagnosis: device calibration, clocks, timed regions -------------------------------------------------------

def _hwmon_dir(dev):
    """sysfs hwmon directory of the torch device (read-only probes: clocks, power, temperatures), or None."""
    try:
        pr = torch.cuda.get_device_properties(dev)
        bdf = '%04x:%02x:%02x.0' % (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id)
        hw = sorted(Path('/sys/bus/pci/devices/%s/hwmon' % bdf).glob('hwmon*'))
        return hw[0] if hw else None
    except Exception:
        return None


def read_clocks(hw):
    """One sample of {sclk_mhz, mclk_mhz, power_w, temp_*_c} from the device's hwmon files (what rocm-smi prints), or {}."""
    out = {}
    if hw is None:
        return out
    try:
        for f in hw.glob('*_input'):
            stem = f.name[:-6]
            try:
                v = float(f.read_text())
            except Exception:
                continue
            lab = hw / (stem + '_label')
            name = lab.read_text().strip() if lab.exists() else stem
            if stem.startswith('freq'):
                out['%s_mhz' % name] = round(v / 1e6)
            elif stem.startswith('power'):
                out['power_w'] = round(v / 1e6)
            elif stem.startswith('temp'):
                out['temp_%s_c' % name] = round(v / 1e3)
    except Exception:
        pass
    return out


class ClockSampler:
    """Samples the hwmon files every `period` seconds on a host thread (under load the files show what the chip actually holds; an idle
    read shows 95 MHz).  Started BEFORE the stabilising warm-up: the thread's first reads cost the GPU ~0.7 % for a few hundred
    milliseconds (measured: the first timed region 6.14 vs 6.09 ms when the sampler started with it); `summary(windows)` keeps the samples
    that fall inside the timed regions."""

    def __init__(self, dev, period=0.04):
        import threading
        self.hw = None if os.environ.get('MNR_BENCH_NO_SAMPLER') else _hwmon_dir(dev)
        self.period, self.samples, self._stop = period, [], threading.Event()
        self._t = threading.Thread(target=self._run, daemon=True)

    def _run(self):
        while not self._stop.is_set():
            c = read_clocks(self.hw)
            if c:
                self.samples.append((time.perf_counter(), c))
            self._stop.wait(self.period)

    def start(self):
        if self.hw is not None:
            self._t.start()
        return self

    def stop(self):
        self._stop.set()
        if self.hw is not None and self._t.is_alive():
            self._t.join()

    def summary(self, windows):
        keep = [c for t, c in self.samples if any(a <= t <= b for a, b in windows)]
        if not keep:
            return None
        out = {'n': len(keep)}
        for k in keep[0]:
            v = [s_[k] for s_ in keep if k in s_]
            out[k] = {'min': min(v), 'mean': round(sum(v) / len(v), 1), 'max': max(v)}
        return out


def xcd_clocks_under_load(step, seconds=1.2):
    """Per-XCD graphics clocks while the step keeps running (amd-smi on a host thread; the hwmon files give one sclk only): a chip whose
    XCDs do not hold the same clock shows here.  None when amd-smi is missing."""
    import shutil
    import subprocess
    import threading
    if shutil.which('amd-smi') is None:
        return None
    res = {}

    def probe():
        try:
            res['txt'] = subprocess.run(['amd-smi', 'metric', '--clock', '--power', '--json'], capture_output=True, text=True, timeout=20).stdout
        except Exception as e:
            res['err'] = str(e)
    th = threading.Thread(target=probe, daemon=True)
    th.start()
    t0 = time.perf_counter()
    while th.is_alive() or time.perf_counter() - t0 < seconds:
        for _ in range(5):
            step()
        torch.cuda.synchronize()
        if time.perf_counter() - t0 > 15:
            break
    th.join(timeout=20)
    try:
        d = json.loads(res['txt'])
        g = d[0] if isinstance(d, list) else (d.get('gpu_data') or [d])[0]
        clk = g.get('clock', {})
        out = {'gfx_mhz': [clk[k]['clk']['value'] if isinstance(clk[k].get('clk'), dict) else clk[k].get('clk') for k in sorted(clk) if k.startswith('gfx_')],
               'mem_mhz': [clk[k]['clk']['value'] if isinstance(clk[k].get('clk'), dict) else clk[k].get('clk') for k in sorted(clk) if k.startswith('mem_')]}
        pw = g.get('power', {})
        sp = pw.get('socket_power')
        out['socket_power_w'] = sp.get('value') if isinstance(sp, dict) else sp
        out['throttle_status'] = pw.get('throttle_status')
        return out
    except Exception as e:
        return {'error': '%s: %s' % (type(e).__name__, e), 'raw': (res.get('txt') or res.get('err') or '')[:200]}


def calibrate(dev):
    """mnr_calibrate: fp32-MFMA rate, L2 -> LDS weight-stream shape, dependent-load latencies, HBM streams, single-wavefront clock
    (csrc/calibrate.hip).  ~30 ms; its 1 GiB scratch is released again."""
    from mega_nerf import _native
    try:
        scr = torch.empty(1 << 30, dtype=torch.uint8, device=dev)
        _native.calibrate(dev, scr)                       # (first call: code load)
        c = _native.calibrate(dev, scr)
        del scr
        return {k: v for k, v in c.items() if k not in ('cu_count', 'nominal_sclk_mhz', 'nominal_mclk_mhz', 'l2_bytes')}, {
            k: c[k] for k in ('cu_count', 'nominal_sclk_mhz', 'nominal_mclk_mhz', 'l2_bytes')}
    except Exception as e:
        return {'error': '%s: %s' % (type(e).__name__, e)}, {}


def rank_devices(dev, dist, world, cal=None):
    """What each rank runs on (answers "did RCCL see N ranks on N GPUs" from the record) and, in short, what its GPU delivered right
    before the timed regions."""
    pr = torch.cuda.get_device_properties(dev)
    me = {'rank': int(os.environ.get('RANK', 0)), 'device': dev.index, 'name': pr.name,
          'pci': '%04x:%02x:%02x.0' % (getattr(pr, 'pci_domain_id', 0), getattr(pr, 'pci_bus_id', 0), getattr(pr, 'pci_device_id', 0)),
          'uuid': str(getattr(pr, 'uuid', '')), 'cus': pr.multi_processor_count}
    if cal and 'error' not in cal:
        me['cal'] = {k: cal.get(k) for k in ('mfma_f32_tflops', 'mfma_wg_ms_median', 'mfma_wg_ms_max', 'sclk_mhz_mfma_chain', 'chase_hbm_ns', 'hbm_read_gbps')}
    if dist is None:
        return {'world': 1, 'backend': None, 'ranks': [me]}
    got = [None] * world
    dist.all_gather_object(got, me)
    return {'world': world, 'backend': dist.get_backend(), 'ranks': got}


def self_launch(args):
    """`python bench.py --gpus N` (N > 1) without a launcher: start the N ranks ourselves, exactly the way the driver does
    (torch.distributed.run, one rank per GPU, rendezvous on 127.0.0.1) and hand back its exit code."""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(('127.0.0.1', 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus), '--master-addr', '127.0.0.1',
           '--master-port', str(port), str(Path(__file__).resolve())] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    env.setdefault('OMP_NUM_THREADS', '1')
    return subprocess.call(cmd, env=env)


# ---- main ------------------------------------------------------------------------------------------------------------

def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=50)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--mode', choices=['eval', 'train'], default='train',
                    help='train: fwd+bwd+Adam step (BASELINE metric: train rays/s); eval: render_rays forward only')
    ap.add_argument('--rays', type=int, default=1024, help='rays per batch (BASELINE: 1024)')
    ap.add_argument('--samples', default='64,128', help='coarse,fine samples per ray (BASELINE: 64,128; reference default 256,512)')
    ap.add_argument('--submodules', type=int, default=0, metavar='S',
                    help='strong scaling: a fixed set of S submodules dealt round-robin to the ranks (0 = one private submodule per rank)')
    ap.add_argument('--layer-dim', type=int, default=256,
                    help='MLP width (256 = the headline Rubble config; 512 = configs/mega-nerf Building: training through the tiled per-layer GEMMs, inference through the wavefront-pair kernel)')
    ap.add_argument('--sh-deg', type=int, default=None,
                    help='spherical-harmonics colour head (BASELINE configs[4], configs/mega-nerf-sh-3/*.yaml: sh_deg 2, pos_dir_dim 0)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-extras', action='store_true', help='skip the N = 1 side measurements and the PSNR check')
    ap.add_argument('--no-diag', dest='no_diag', action='store_true',
                    help='skip the device calibration probes around the timed regions (profiling passes: keeps their kernels out of the trace)')
    ap.add_argument('--no-config-sweep', action='store_true',
                    help='skip the compact lines of the other BASELINE configs (8-cell set, container, W=512, SH) in the default run')
    ap.add_argument('--only-split-extras', action='store_true',
                    help='of the side measurements keep the split-precision ones only (profiling runs of the k_mlp_*_h2 kernels)')
    ap.add_argument('--container', type=int, default=0, metavar='N',
                    help='eval mode only: render through a merged N-cell container (MegaNeRF router, boundary_margin 1.15) '
                         'instead of one submodule -- the "8-submodule Rubble" evaluation shape on ONE GPU')
    return ap.parse_args(argv)


def main():
    args = parse_args()
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        raise SystemExit(self_launch(args))
    rank = int(os.environ.get('RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X: the hot path has no CPU fallback')
    if os.environ.get('MNR_BENCH_SHARE_GPU'):          # diagnostics: several ranks on ONE GPU (exercises the N > 1 code path on a 1-GPU box)
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    dist = None
    if world > 1 or os.environ.get('MNR_BENCH_FORCE_DIST'):      # (FORCE_DIST: the RCCL branch with ONE rank -- what a 1-GPU box can execute of it)
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29541')
        os.environ.setdefault('RANK', '0'), os.environ.setdefault('WORLD_SIZE', '1')
        if os.environ.get('MNR_BENCH_SHARE_GPU'):
            dist.init_process_group('gloo')                       # (RCCL refuses two ranks on one device)
        else:
            dist.init_process_group('nccl', device_id=dev)        # RCCL on ROCm, communicator bound to this rank's GPU
    if args.gpus != world:
        raise SystemExit('--gpus %d does not match WORLD_SIZE %d (start it as `python bench.py --gpus N` or under torch.distributed.run with N ranks)' % (args.gpus, world))
    line = run_config(args, rank, world, dev, dist)
    headline = (world == 1 and args.mode == 'train' and not args.submodules and not args.container and args.layer_dim == 256 and
                args.sh_deg is None and args.samples == '64,128' and args.rays == 1024)
    if rank == 0 and headline and not args.no_extras and not args.no_config_sweep:
        t0 = time.perf_counter()
        line['baseline_configs'] = config_sweep(args, dev)
        set8 = next((v for k, v in line['baseline_configs'].items() if k.startswith('configs[2] Rubble 8 submodules')), None)
        if set8 and 'error' not in set8:
            # the N = 1 point of the STRONG-scaling curve BASELINE.json's metric is quoted on ("Rubble 8-submodule"): the fixed 8-cell set on one
            # GPU (`--gpus N --submodules 8` deals the same set to N ranks; there is no exchange, so N = 8 is one cell's step)
            line['strong_scaling_n1'] = {'flags': '--submodules 8', 'submodules': 8, 'ms_per_step_of_the_set': set8['ms_per_step'],
                                         'rays_per_sec': set8['rays_per_sec'], 'frac_of_f32_mfma_peak': set8['frac'], 'steps': set8['steps']}
        line['runner_loop'] = runner_loop(args, dev, line['value'])
        line['joint_cells_loop'] = joint_cells_loop(args, dev)
        line['baseline_configs']['_seconds'] = round(time.perf_counter() - t0, 1)
    if rank == 0:
        if 'diag' in line:
            line['diag'] = line.pop('diag')          # last key of the line: the driver's record keeps the tail
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.destroy_process_group()


# BASELINE.json `configs` beyond the headline one (configs[1]), each as a compact line of the default run: same code path as the
# flag combination named in `flags`, short timed region, no CPU baseline, no side measurements
# (third field: timed steps -- every entry times >= 0.3 s)
SWEEP = [
    ('configs[2] Rubble 8 submodules, ONE GPU trains the whole set (one mnr_train_step call per iteration)', ['--submodules', '8', '--mode', 'train'], 10),
    ('configs[2] Rubble merged 8-cell container, routed eval', ['--container', '8', '--mode', 'eval'], 90),
    ('configs[3] Building-shaped cell (fg 8x512), train', ['--layer-dim', '512', '--mode', 'train'], 15),
    ('configs[3] Building-shaped cell (fg 8x512), eval', ['--layer-dim', '512', '--mode', 'eval'], 50),
    # 25 Building cells on 8 GPUs = 4,3,3,...: the busiest rank's four 512-wide cells in ONE plan (what tools/train_cells.py runs per rank)
    ('configs[3] Building: four 512-wide cells of one rank in one plan (25 cells on 8 GPUs = 4,3,3,..), train', ['--layer-dim', '512', '--submodules', '4', '--mode', 'train'], 4),
    ('configs[3] Building merged 25-cell container of 512-wide cells, routed eval', ['--layer-dim', '512', '--container', '25', '--mode', 'eval'], 30),
    ('configs[4] Sci-Art-shaped cell (sh_deg 2, pos_dir_dim 0), train', ['--sh-deg', '2', '--mode', 'train'], 50),
    ('configs[4] Sci-Art-shaped cell (sh_deg 2, pos_dir_dim 0), eval', ['--sh-deg', '2', '--mode', 'eval'], 150),
    # BASELINE.json words configs[4] as "SH-degree-3"; the reference's config files say sh_deg 2.  Degree 3 (48 colour
    # coefficients) has its own pair of the one-call step / render
    ('configs[4] as worded in BASELINE.json: sh_deg 3, train', ['--sh-deg', '3', '--mode', 'train'], 50),
    ('configs[4] as worded in BASELINE.json: sh_deg 3, eval', ['--sh-deg', '3', '--mode', 'eval'], 150),
]


def config_sweep(args, dev):
    import gc
    out = {}
    # The headline's cpu_baseline leg ran torch on every host thread the process may use; its OpenMP workers keep spinning for a while after
    # every later CPU-side tensor op, next to the one Python thread that enqueues the launches -- the routed-container entry (36 launches
    # per 3.5 ms render) then measured 3.69 ms here against 3.56 ms on its own.  The GPU path needs one host thread.
    host_threads = torch.get_num_threads()
    torch.set_num_threads(1)
    for name, flags, steps in SWEEP:
        a = parse_args(flags + ['--steps', str(steps), '--warmup', '3', '--no-cpu-baseline', '--no-extras', '--rays', str(args.rays), '--samples', args.samples])
        a.no_diag = True                      # (no device calibration per side line; their three timed regions are reported)
        t0 = time.perf_counter()
        try:
            ln = run_config(a, 0, 1, dev, None)
            r = ln.get('roofline') or {}
            out[name] = {'flags': ' '.join(flags), 'ms_per_step': round(ln['ms_per_step'], 4), 'rays_per_sec': round(ln['value'], 1),
                         'steps': a.steps, 'frac': r.get('frac'), 'frac_of': (r.get('kernel') or '').split(' (')[0], 'peak_tflops': r.get('peak'),
                         'achieved_tflops': r.get('achieved'), 'workload': ln['config']['workload'],
                         'seconds': round(time.perf_counter() - t0, 1)}
            for k in ('step_spans_ms',):
                if k in ln:
                    out[name][k] = ln[k]
            out[name]['host_enqueue_ms_per_step'] = (ln.get('host') or {}).get('host_enqueue_ms_per_step')
            out[name]['host_blocked_ms_per_step'] = (ln.get('host') or {}).get('host_blocked_ms_per_step')
            out[name]['regions_ms_per_step'] = ((ln.get('diag') or {}).get('timing') or {}).get('regions_ms_per_step')
            if 'routed_rows_per_step' in r:
                out[name]['routed_rows_per_step'] = r['routed_rows_per_step']
        except Exception as e:                      # a side line must never take the headline down
            out[name] = {'flags': ' '.join(flags), 'error': '%s: %s' % (type(e).__name__, e)}
        gc.collect()
        torch.cuda.empty_cache()
    for mode in ('train', 'eval'):
        name = 'configs[0] configs/nerf-shaped model (cascade, layer_dim 2048, no appearance, no background), %s' % mode
        try:
            out[name] = config0_line(args, dev, mode)
        except Exception as e:
            out[name] = {'flags': 'configs/nerf/*.yaml', 'error': '%s: %s' % (type(e).__name__, e)}
        gc.collect()
        torch.cuda.empty_cache()
    torch.set_num_threads(host_threads)
    return out


def config0_line(args, dev, mode):
    """BASELINE.json configs[0] on the MI355X: `configs/nerf/*.yaml` (use_cascade, layer_dim 2048, appearance_dim 0, no_bg_nerf; /root/reference's
    configs/nerf/rubble.yaml:1-4) at the benchmark's 1024 rays x (64 + 128) samples.  A cascade evaluates the coarse model on Nc samples and
    the fine model on the sorted Nc + Nf: 64 + 192 = 256 MLP rows per ray.  2048-wide layers run on the tiled per-layer GEMMs
    (k_tgemm / k_wgrad2 jobs); whole-step figure (wall clock incl. the render stages) against the fp32-MFMA peak."""
    import synthetic_scene as S
    from mega_nerf import ray_utils
    from mega_nerf.models.model_utils import get_nerf
    from mega_nerf.opts import get_opts_base
    from mega_nerf.rendering import render_rays_async
    from mega_nerf.training import TrainStep
    Nc, Nf = [int(v) for v in args.samples.split(',')]
    hp = get_opts_base().parse_args(['--coarse_samples', str(Nc), '--fine_samples', str(Nf), '--layer_dim', '2048', '--appearance_dim', '0',
                                     '--use_cascade', '--no_bg_nerf'])
    torch.manual_seed(7)
    nerf = get_nerf(hp, 0).to(dev)
    s = S.SCENE
    d = ray_utils.get_ray_directions(s['W'], s['H'], s['fx'], s['fy'], s['cx'], s['cy'], True, dev)
    all_rays = ray_utils.get_rays(d, torch.from_numpy(s['c2w']).to(dev), s['near'], 2.0, s['ray_altitude_range']).view(-1, 8)     # no background: far = 2 (runner.py:81-82)
    g = torch.Generator(device='cpu').manual_seed(42)
    rays = all_rays[torch.randperm(all_rays.shape[0], generator=g)[:args.rays].to(dev)].contiguous()
    target = torch.rand(args.rays, 3, generator=g).to(dev)
    mac = sum(p.numel() for k_, p in nerf.fine.named_parameters() if k_.endswith('weight'))
    steps = 6 if mode == 'train' else 20
    if mode == 'train':
        nerf.train()
        ts = TrainStep(nerf, None, hp, None, None)
        fn = lambda: ts(rays, None, target)                                                           # noqa: E731
    else:
        nerf.eval()

        def fn():
            with torch.no_grad():
                render_rays_async(nerf, None, rays, None, hp, None, None, True, False, True)
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    fl = 2.0 * mac * args.rays * (Nc + Nc + Nf) * (3 if mode == 'train' else 1)
    return {'flags': 'configs/nerf/*.yaml: use_cascade, layer_dim 2048, appearance_dim 0, no_bg_nerf', 'ms_per_step': round(dt * 1e3, 4),
            'rays_per_sec': round(args.rays / dt, 1), 'steps': steps, 'frac': round(fl / dt / 1e12 / PEAK_F32_MFMA_TFLOPS, 4),
            'frac_of': 'whole step (k_tgemm / k_wgrad2 jobs + render stages), wall clock', 'peak_tflops': PEAK_F32_MFMA_TFLOPS,
            'achieved_tflops': round(fl / dt / 1e12, 2), 'algorithmic_gflop_per_step': round(fl / 1e9, 1),
            'workload': 'cascade of two 8x2048 NeRFs, %d rays x (%d coarse + %d fine-model) rows, fg only' % (args.rays, Nc, Nc + Nf)}


def joint_cells_loop(args, dev, n_cells=4):
    """Rays/s of a rank that owns several cells, trained the way tools/train_cells.py trains them: every cell its own, unchanged
    Runner.train() loop on a host thread, the cells' iterations meeting in ONE mnr_train_step per iteration (training.JointCells) -- against
    the bare multi-cell step on the last batches (what `--submodules N` times).  Same synthetic dataset for every cell (throughput does not
    depend on the pixels); timed from iteration 20 of cell 0 to its last one."""
    import contextlib
    import io
    import shutil
    import tempfile
    import numpy as np
    import synthetic_scene as S
    from PIL import Image
    from mega_nerf.opts import get_opts_base
    from mega_nerf.runner import Runner
    from mega_nerf.training import JointCells
    tmp = Path(tempfile.mkdtemp(prefix='mnr_bench_joint_'))
    try:
        data, sc = tmp / 'data', S.SCENE
        rng = np.random.default_rng(7)
        torch.save({'origin_drb': torch.zeros(3), 'pose_scale_factor': 1.0}, _mkdir(data) / 'coordinates.pt')
        for i, split in enumerate(('train', 'train', 'val')):
            c2w = torch.from_numpy(sc['c2w'].copy())
            c2w[:, 3] += torch.tensor([0.0, 0.01 * i, -0.01 * i])
            Image.fromarray(rng.integers(0, 256, (sc['H'], sc['W'], 3), dtype=np.uint8)).save(_mkdir(data / split / 'rgbs') / ('%06d.png' % i))
            torch.save({'W': sc['W'], 'H': sc['H'], 'intrinsics': torch.tensor([sc['fx'], sc['fy'], sc['cx'], sc['cy']]), 'c2w': c2w},
                       _mkdir(data / split / 'metadata') / ('%06d.pt' % i))
        p = get_opts_base()
        p.add_argument('--exp_name', type=str, required=True)
        p.add_argument('--dataset_path', type=str, required=True)
        iters, first = 20 + args.steps, 20
        runners, marks = [], {}
        with contextlib.redirect_stdout(io.StringIO()):
            joint = JointCells(n_cells)
            for c in range(n_cells):
                hp = p.parse_args(['--dataset_path', str(data), '--exp_name', str(tmp / ('exp%d' % c)), '--coarse_samples', '64', '--fine_samples', '128',
                                   '--near', str(sc['near']), '--ray_altitude_range'] + [str(v) for v in sc['ray_altitude_range']] +
                                  ['--val_scale_factor', '8', '--batch_size', str(args.rays), '--train_iterations', str(iters), '--random_seed', str(42 + c)])
                r = Runner(hp)
                r.sphere_center = torch.from_numpy(sc['sphere_center']).to(dev)
                r.sphere_radius = torch.from_numpy(sc['sphere_radius']).to(dev)
                r.trainer_factory = joint.member(c)
                r._write_final_metrics = lambda *a, **k: None          # (no validation render behind the loops: this times training)
                r._run_validation = lambda *a, **k: {'val/psnr': 0.0, 'val/ssim': 0.0}
                runners.append(r)

            def hook(it):
                if it in (first, iters):
                    torch.cuda.synchronize()
                    marks[it] = time.perf_counter()
            runners[0].iteration_hook = hook
            joint.run([r.train for r in runners])
        dt = (marks[iters] - marks[first]) / (iters - first)
        bare = None
        if joint.plan is not None:
            fs, batches = joint.plan, list(joint.plan._keep[:n_cells])
            for _ in range(3):
                fs(batches)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(iters - first):
                fs(batches)
            torch.cuda.synchronize()
            bare = (time.perf_counter() - t1) / (iters - first)
        return {'cells': n_cells, 'rays_per_sec': round(n_cells * args.rays / dt, 1), 'ms_per_joint_iteration': round(dt * 1e3, 4),
                'joint_steps': joint.joint_steps, 'cell_by_cell_iterations': joint.separate_steps,
                'bare_multi_cell_step_on_its_last_batches_ms': round(bare * 1e3, 4) if bare else None,
                'fraction_of_bare_step': round(bare / dt, 4) if bare else None,
                'what': 'training.JointCells: %d cells, each its own Runner.train() loop on a host thread, one mnr_train_step per iteration for all of them '
                        '(tools/train_cells.py on a rank that owns several cells)' % n_cells}
    except Exception as e:
        return {'error': '%s: %s' % (type(e).__name__, e)}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def runner_loop(args, dev, value):
    """Rays/s of the DROP-IN trainer: mega_nerf.runner.Runner.train() (the loop train.py runs, reference runner.py:244-277) over a
    device-resident dataset of the benchmark's own camera (three 400 x 400 images of the SURVEY 8(d) pose, the benchmark's ellipsoid:
    the same ~13 % of rays with a background segment as the headline batch), timed from iteration 20 to the last one (a hook the Runner
    calls after every iteration synchronises at the two ends): dataset.batches() gathers + the one-call step + ExponentialLR + the
    periodic health check.  Image contents are noise: throughput does not depend on them."""
    import contextlib
    import io
    import shutil
    import tempfile
    import numpy as np
    import synthetic_scene as S
    from PIL import Image
    from mega_nerf.opts import get_opts_base
    from mega_nerf.runner import Runner
    tmp = Path(tempfile.mkdtemp(prefix='mnr_bench_'))
    try:
        data, sc = tmp / 'data', S.SCENE
        rng = np.random.default_rng(7)
        torch.save({'origin_drb': torch.zeros(3), 'pose_scale_factor': 1.0}, _mkdir(data) / 'coordinates.pt')
        for i, split in enumerate(('train', 'train', 'val')):
            c2w = torch.from_numpy(sc['c2w'].copy())
            c2w[:, 3] += torch.tensor([0.0, 0.01 * i, -0.01 * i])              # (distinct camera centres: the Runner derives its bounds from them)
            Image.fromarray(rng.integers(0, 256, (sc['H'], sc['W'], 3), dtype=np.uint8)).save(_mkdir(data / split / 'rgbs') / ('%06d.png' % i))
            torch.save({'W': sc['W'], 'H': sc['H'], 'intrinsics': torch.tensor([sc['fx'], sc['fy'], sc['cx'], sc['cy']]), 'c2w': c2w},
                       _mkdir(data / split / 'metadata') / ('%06d.pt' % i))
        p = get_opts_base()
        p.add_argument('--exp_name', type=str, required=True)
        p.add_argument('--dataset_path', type=str, required=True)
        iters, first = 20 + args.steps, 20
        hp = p.parse_args(['--dataset_path', str(data), '--exp_name', str(tmp / 'exp'), '--coarse_samples', '64', '--fine_samples', '128',
                           '--near', str(sc['near']), '--ray_altitude_range'] + [str(v) for v in sc['ray_altitude_range']] +
                          ['--val_scale_factor', '8', '--batch_size', str(args.rays), '--train_iterations', str(iters)])
        with contextlib.redirect_stdout(io.StringIO()):
            r = Runner(hp)
            r.sphere_center = torch.from_numpy(sc['sphere_center']).to(dev)       # the benchmark's ellipsoid instead of the one three
            r.sphere_radius = torch.from_numpy(sc['sphere_radius']).to(dev)       # near-identical cameras would span
            marks = {}

            def hook(it):
                if it in (first, iters):
                    torch.cuda.synchronize()
                    marks[it] = time.perf_counter()
                    if it == iters:
                        marks['n_bg'] = int(r.trainer.fused.n_bg[0]) if r.trainer is not None and r.trainer.fused is not None else -1
            r.iteration_hook = hook
            r.train()
        dt = (marks[iters] - marks[first]) / (iters - first)
        fused = r.trainer is not None and r.trainer.fused is not None
        bare = None
        if fused:
            # the bare step on a batch of THIS dataset (random rays of the whole image carry more background segments than the headline
            # batch: the loop's own cost is the difference to this figure, not to `value`)
            fs, batch = r.trainer.fused, r.trainer.fused._keep[0]
            for _ in range(3):
                fs([batch])
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(iters - first):
                fs([batch])
            torch.cuda.synchronize()
            bare = (time.perf_counter() - t1) / (iters - first)
        return {'rays_per_sec': round(args.rays / dt, 1), 'ms_per_iteration': round(dt * 1e3, 4), 'iterations_timed': iters - first,
                'fraction_of_value': round(args.rays / dt / value, 4), 'one_call_step': bool(fused), 'bg_rays_in_last_batch': marks.get('n_bg'),
                'bare_step_on_its_last_batch_ms': round(bare * 1e3, 4) if bare else None,
                'fraction_of_bare_step_on_its_last_batch': round(bare / dt, 4) if bare else None,
                'what': 'mega_nerf.runner.Runner.train() on a device-resident MemoryDataset of the benchmark camera (%d pixels), batch %d: '
                        'row selections gathered inside mnr_train_step (no torch kernel per iteration) + ExponentialLR + health check every 100 iterations' % (
                            int(2.5 * sc['H'] * sc['W']), args.rays)}
    except Exception as e:
        return {'error': '%s: %s' % (type(e).__name__, e)}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def _mkdir(p: Path) -> Path:
    p.mkdir(parents=True, exist_ok=True)
    return p


def run_config(args, rank, world, dev, dist):
    """One measured configuration -> the JSON line (rank 0; None elsewhere)."""
    import synthetic_scene as S                    # seeded scene / weight generator (pure numpy; no oracle code)
    from mega_nerf import ray_utils, rendering
    from mega_nerf.distributed import assign_submodules
    from mega_nerf.opts import get_opts_base
    from mega_nerf.rendering import render_rays_async
    from mega_nerf.training import TrainStep

    Nc, Nf = [int(v) for v in args.samples.split(',')]
    # opts.py defaults = configs/mega-nerf (8x256, 12/4 frequency bands, 48-d appearance) at the benchmark's samples per ray
    hp = get_opts_base().parse_args(['--coarse_samples', str(Nc), '--fine_samples', str(Nf), '--layer_dim', str(args.layer_dim)] +
                                    (['--sh_deg', str(args.sh_deg), '--pos_dir_dim', '0'] if args.sh_deg is not None else []))
    wide = args.layer_dim != 256 or args.sh_deg is not None          # side measurement of the other architectures: whole-step figures only
    s = S.SCENE
    sc, sr = torch.from_numpy(s['sphere_center']).to(dev), torch.from_numpy(s['sphere_radius']).to(dev)
    d = ray_utils.get_ray_directions(s['W'], s['H'], s['fx'], s['fy'], s['cx'], s['cy'], True, dev)
    all_rays = ray_utils.get_rays(d, torch.from_numpy(s['c2w']).to(dev), s['near'], s['far'], s['ray_altitude_range']).view(-1, 8)

    def make_batch(seed, n_rays):
        """synthetic batch: rays of the 400x400 camera by seeded permutation, ~13 % bg rays"""
        g = torch.Generator(device='cpu').manual_seed(seed)
        sel = torch.randperm(all_rays.shape[0], generator=g)[:n_rays].to(dev)
        return (all_rays[sel].contiguous(), torch.randint(0, s['appearance_count'], (n_rays,), generator=g).float().to(dev),
                torch.rand(n_rays, 3, generator=g).to(dev))

    # the cells this rank owns: one private submodule (weak) or its share of the fixed set (strong)
    cells = assign_submodules(args.submodules, world)[rank] if args.submodules else [rank]
    total_cells = args.submodules if args.submodules else world
    work = []
    for c in cells:
        (fg, fcfg, fw), (bg, bcfg, bw) = build_models(hp, dev, 1000 * (c + 1), args.layer_dim)
        work.append(dict(fg=fg, bg=bg, batch=make_batch(42 + c, args.rays), fcfg=fcfg, bcfg=bcfg, fw=fw, bw=bw))
    if args.container:
        assert args.mode == 'eval' and not args.submodules, '--container is a single-GPU evaluation shape (routed containers are inference-only)'
        from mega_nerf.models.mega_nerf import MegaNeRF
        n = args.container
        g0 = max(1, int(round(n ** 0.5)) if int(round(n ** 0.5)) ** 2 == n else 2)
        g1 = n // g0
        assert g0 * g1 == n, '--container must factor into a grid'
        cent = torch.stack([torch.zeros(n), torch.linspace(-.45, .45, g0).repeat_interleave(g1), torch.linspace(-.45, .45, g1).repeat(g0)], 1)
        sub = [build_models(hp, dev, 1000 * (rank + 1) + 7 * j, args.layer_dim) for j in range(n)]
        work[0]['fg'] = MegaNeRF([c[0][0] for c in sub], cent, hp.boundary_margin, False, False).to(dev)
        work[0]['bg'] = MegaNeRF([c[1][0] for c in sub], cent, hp.boundary_margin, True, False).to(dev)
        for k in ('fg', 'bg'):               # device-side tally of the rows the router hands to cells (a row near a boundary goes to two):
            work[0][k].routed_rows = None    # switched on for ONE render behind the timed region (three small kernels per routed evaluation)
        work[0]['cells_np'] = dict(cent=cent.numpy(), fg=[c[0][2] for c in sub], bg=[c[1][2] for c in sub], fcfg=sub[0][0][1], bcfg=sub[0][1][1])
        hp.container_path = 'bench'              # background points carry their world position for the router (quirk Q15)

    steppers, trainers = [], []
    fused = None
    if args.mode == 'train' and (not wide or (args.layer_dim == 512 and args.sh_deg is None)):
        # every cell this rank owns goes through ONE mnr_train_step call per step (csrc/step.hip): 12 kernel launches + a memset for the
        # whole iteration, the cells' rows side by side in the MLP launches
        from mega_nerf.training import FusedTrainStep, fused_step_supported
        for w in work:
            w['fg'].train(), w['bg'].train()
        if all(fused_step_supported(w['fg'], w['bg'], hp, args.rays) for w in work):
            fused = FusedTrainStep([(w['fg'], w['bg']) for w in work], hp, sc, sr, args.rays)
            batches = [w['batch'] for w in work]

            def fused_step():
                loss, n_bg, err = fused(batches)
                return loss[-1], n_bg[-1], err[-1]
            steppers.append(fused_step)
    for w in (work if fused is None else []):
        if args.mode == 'train':
            w['fg'].train(), w['bg'].train()
            ts = TrainStep(w['fg'], w['bg'], hp, sc, sr)
            trainers.append(ts)
            steppers.append(lambda ts=ts, b=w['batch']: ts(*b))
        else:
            w['fg'].eval(), w['bg'].eval()

            def ev_step(w=w):
                with torch.no_grad():
                    return render_rays_async(w['fg'], w['bg'], w['batch'][0], w['batch'][1], hp, sc, sr, True, False, True)
            steppers.append(ev_step)

    def step():
        out = None
        for f in steppers:
            out = f()
        return out

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    diagnose = rank == 0 and not getattr(args, 'no_diag', False)
    # (every rank calibrates its own GPU before the regions -- one slow device among N sets the max-over-ranks time --; rank 0 also after)
    cal_before, dev_info = calibrate(dev) if not getattr(args, 'no_diag', False) else (None, {})
    # Clock-stabilised warm-up on top of the contract's `--warmup` steps: blocks of steps until two consecutive blocks agree to 1 % (cap
    # 1 s).  A fresh process starts the timed region on a chip that idled at 95 MHz a few milliseconds earlier.
    # kernel-level timing of the MLP / weight-gradient launches with HIP events recorded on the launch stream inside the timed regions: one
    # slot per step of a region, re-used by the next region (read out between regions); switched on BEFORE the stabilising steps, which
    # then also pay every event's first-use cost (a fresh event's first record is ~1.5 us dearer: 1 % of a region when the events were made
    # right in front of it)
    REGIONS = 3
    rendering.KERNEL_EVENTS = None
    if fused is not None:
        fused.profile(args.steps)
    clocks = ClockSampler(dev).start()
    stab, t_stab0, blk = [], time.perf_counter(), max(2, min(10, args.steps))
    while True:
        t = time.perf_counter()
        for _ in range(blk):
            step()
        torch.cuda.synchronize()
        stab.append((time.perf_counter() - t) / blk * 1e3)
        done = (blk * len(stab) >= min(args.steps, 60) or fused is None) and time.perf_counter() - t_stab0 >= 0.3     # (every profiling slot touched once; the sampler's first reads behind us)
        if (done and len(stab) >= 2 and abs(stab[-1] - stab[-2]) <= 0.01 * stab[-1]) or time.perf_counter() - t_stab0 > 1.0 or len(stab) >= 50:
            break
    rendering.KERNEL_EVENTS = ev = []
    ms0 = torch.cuda.memory_stats(dev)
    wait0 = sum(getattr(t_, 'host_wait_s', 0.0) for t_ in trainers)
    region_s, enq_s, span_ms, windows = [], [], {}, []
    for _r in range(REGIONS):
        # EXACTLY `--steps` steps per region, a barrier + synchronize on both sides; three disjoint regions, `ms_per_step` = their median
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            out = step()
        enq_s.append(time.perf_counter() - t0)     # host time to ENQUEUE the timed steps (diagnostic: host-bound when ~ total)
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        region_s.append(time.perf_counter() - t0)
        windows.append((t0, time.perf_counter()))
        if fused is not None:             # this region's kernel spans (the slots are re-used by the next region)
            for i in range(args.steps):
                for k, v in fused.kernel_times(i).items():
                    span_ms.setdefault(k, []).append(v)
    clocks.stop()
    wait1 = sum(getattr(t_, 'host_wait_s', 0.0) for t_ in trainers)
    cal_after = calibrate(dev)[0] if diagnose else None
    rendering.KERNEL_EVENTS = None
    xcd = xcd_clocks_under_load(step) if (diagnose and not args.no_extras) else None
    if args.container:      # the router's device-side tally: the batch is fixed, so one more render of it counts what every timed one routed
        for k in ('fg', 'bg'):
            work[0][k].routed_rows = torch.zeros((), device=dev, dtype=torch.int64)
        step()
        routed_timed = (int(work[0]['fg'].routed_rows) * args.steps, int(work[0]['bg'].routed_rows) * args.steps)
        work[0]['fg'].routed_rows = work[0]['bg'].routed_rows = None
    if args.mode == 'eval' and not ev and not args.container:
        # the timed steps went through mnr_render_fwd (six launches, no Python between them): take the kernel-level timings of
        # the MLP launches -- the same kernel over the same rows -- from the stage-by-stage sequencing of the same render
        rendering.FUSED_RENDER, rendering.KERNEL_EVENTS = False, ev
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()
        rendering.FUSED_RENDER, rendering.KERNEL_EVENTS = True, None
    span_region = []
    if fused is not None:
        for r_ in range(REGIONS):         # the MLP forward's two launches, region by region (does a slow region show in the kernel, or around it?)
            sl = slice(r_ * args.steps, (r_ + 1) * args.steps)
            span_region.append(round((sum(span_ms['fwd_c'][sl]) + sum(span_ms['fwd_f'][sl])) / (2 * args.steps), 4))
        fused.profile(0)
    ms1 = torch.cuda.memory_stats(dev)
    # max over ranks, region by region; the reported time is the MEDIAN region
    tmax = torch.tensor(region_s, device=dev, dtype=torch.float64)
    if dist is not None:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    region_s = [float(v) for v in tmax]
    order = sorted(range(REGIONS), key=lambda i_: region_s[i_])
    dt = region_s[order[REGIONS // 2]]
    t_enq = enq_s[order[REGIONS // 2]]
    timing = {'regions_ms_per_step': [round(v / args.steps * 1e3, 4) for v in region_s], 'min': round(min(region_s) / args.steps * 1e3, 4),
              'median': round(dt / args.steps * 1e3, 4), 'max': round(max(region_s) / args.steps * 1e3, 4),
              'spread': round((max(region_s) - min(region_s)) / dt, 4),
              'stabilise_blocks_ms_per_step': [round(v, 4) for v in stab], 'stabilise_steps': blk * len(stab)}
    if span_region:
        timing['fwd_launch_ms_by_region'] = span_region
    blocked = (wait1 - wait0) / (REGIONS * args.steps) * 1e3          # stage-by-stage trainer: the wait for the forward's three scalars
    host_diag = {'host_enqueue_ms_per_step': round(t_enq / args.steps * 1e3 - blocked, 3),
                 'device_mallocs_in_timed_region': int(ms1.get('num_device_alloc', 0) - ms0.get('num_device_alloc', 0)),
                 'alloc_retries_in_timed_region': int(ms1.get('num_alloc_retries', 0) - ms0.get('num_alloc_retries', 0)),
                 'reserved_gb': round(ms1.get('reserved_bytes.all.current', 0) / 1e9, 1)}
    if blocked > 0:
        host_diag['host_blocked_ms_per_step'] = round(blocked, 3)     # (not part of host_enqueue_ms_per_step)
    n_bg = int(out[1]) if out[1] is not None else -1          # background rays in the (last cell's) batch
    rank_info = rank_devices(dev, dist, world, cal_before)
    if not diagnose:
        cal_before = None

    # eval metric all-reduce (packed [sum_psnr, count]) -- the only collective of the path.  The targets of the
    # throughput batches are random colours, so this number only exercises the reduction; the PSNR that means something is
    # the student-vs-teacher check below.
    with torch.no_grad():
        w = work[0]
        res = render_rays_async(w['fg'].eval(), w['bg'].eval(), w['batch'][0], w['batch'][1], hp, sc, sr, True, False, True)[0]
        packed = torch.stack([-10 * torch.log10(torch.mean((res['rgb_fine'] - w['batch'][2]) ** 2)), torch.ones((), device=dev)]).double()
    if dist is not None:
        dist.all_reduce(packed)
    metric_reduce_check = float(packed[0] / packed[1])

    extras = {}
    if rank == 0 and world == 1 and args.submodules and fused is not None and not args.no_extras:
        # the same cell set through the opt-in split-precision step (one mnr_train_step call for all cells): own dtype, not `value`
        from mega_nerf.training import FusedTrainStep
        fs = FusedTrainStep([(w['fg'], w['bg']) for w in work], hp, sc, sr, args.rays, split_precision=True)
        bs = [w['batch'] for w in work]
        for _ in range(3):
            fs(bs)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            fs(bs)
        torch.cuda.synchronize()
        t_sp = (time.perf_counter() - t1) / args.steps
        extras['train_split_precision'] = {
            'dtype': 'forward, data-gradient chain and weight gradients: f16 hi/lo split operands, f32 accumulate (opt-in; the f32 step is `value`)',
            'ms_per_step': t_sp * 1e3, 'rays_per_sec': args.rays * len(work) / t_sp}
        del fs
    if rank == 0 and world == 1 and args.container and args.mode == 'eval' and not args.no_extras and not wide:
        # the routed container with every cell's rows on the opt-in split-precision kernel (mnr_mlp_forward_cells_h2): own dtype, not `value`
        rendering.SPLIT_PRECISION = True
        try:
            for _ in range(3):
                step()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(args.steps):
                step()
            torch.cuda.synchronize()
            extras['eval_split_precision'] = {
                'dtype': 'f16 hi/lo split operands, 3 x v_mfma_f32_16x16x32_f16 per layer, f32 accumulate (opt-in; fp32 kernels are the default)',
                'rays_per_sec': args.rays * args.steps / (time.perf_counter() - t1)}
        finally:
            rendering.SPLIT_PRECISION = False
    if rank == 0 and world == 1 and not args.no_extras and not args.container and not args.submodules and (Nc, Nf) == (64, 128) and not wide:
        w = work[0]
        fgm, bgm = w['fg'], w['bg']

        def timed(fn, reps, warm=2):
            for _ in range(warm):
                fn()
            torch.cuda.synchronize()
            t = time.perf_counter()
            for _ in range(reps):
                fn()
            torch.cuda.synchronize()
            return (time.perf_counter() - t) / reps

        def ev_fn(batch, hpx):
            def f():
                with torch.no_grad():
                    render_rays_async(fgm, bgm, batch[0], batch[1], hpx, sc, sr, True, False, True)
            return f

        fgm.eval(), bgm.eval()
        if args.mode == 'train':
            extras['eval_rays_per_sec_per_gpu'] = args.rays / timed(ev_fn(w['batch'], hp), args.steps, 3)
        big = make_batch(4242, 65536)                                   # image_pixel_batch_size (opts.py:75)
        t_big = timed(ev_fn(big, hp), 3, 1)
        extras['eval_rays_per_sec_65536_ray_batches'] = 65536 / t_big
        with torch.no_grad():
            nbg_big = int(render_rays_async(fgm, bgm, big[0], big[1], hp, sc, sr, True, False, True)[1])
        # whole step (every kernel of render_rays, wall clock) against the MFMA peak: at this batch size the launches are
        # ~40 waves of workgroups, so the partial last wave that costs the 1024-ray launches ~15 % is amortised
        fl_big = 65536 * (Nc + Nf) * FG_FLOP_PER_SAMPLE + nbg_big * (Nc // 2 + Nf // 2) * BG_FLOP_PER_SAMPLE
        extras['eval_65536_ray_batches_whole_step'] = {'tflops': round(fl_big / t_big / 1e12, 1),
                                                       'frac_of_f32_mfma_peak': round(fl_big / t_big / 1e12 / PEAK_F32_MFMA_TFLOPS, 4),
                                                       'bg_rays': nbg_big}
        # opt-in split-precision inference (csrc/mlp_fwd_h2.hip): NOT part of `value` -- its own arithmetic, its own peak
        rendering.SPLIT_PRECISION = True
        try:
            t_sp = timed(ev_fn(w['batch'], hp), args.steps, 3)
            t_sp_big = timed(ev_fn(big, hp), 3, 1)
            with torch.no_grad():
                a_ = render_rays_async(fgm, bgm, w['batch'][0], w['batch'][1], hp, sc, sr, True, False, True)[0]['rgb_fine']
            rendering.SPLIT_PRECISION = False
            with torch.no_grad():
                b_ = render_rays_async(fgm, bgm, w['batch'][0], w['batch'][1], hp, sc, sr, True, False, True)[0]['rgb_fine']
            extras['eval_split_precision'] = {
                'dtype': 'f16 hi/lo split operands, 3 x v_mfma_f32_16x16x32_f16 per layer, f32 accumulate (opt-in; fp32 kernels are the default)',
                'rays_per_sec': args.rays / t_sp, 'rays_per_sec_65536_ray_batches': 65536 / t_sp_big,
                'whole_step_65536': {'f32_equivalent_tflops': round(fl_big / t_sp_big / 1e12, 1),
                                     'issued_f16_mfma_tflops': round(3 * fl_big / t_sp_big / 1e12, 1), 'peak_f16_mfma_tflops': 2500.0,
                                     'frac_of_f16_mfma_peak_issued': round(3 * fl_big / t_sp_big / 1e12 / 2500.0, 4)},
                # (a render is discontinuous where a run of coarse bins has zero probability -- tests/test_gpu_parity_extra.py --, so
                # after training steps a handful of rays may jump between the two kernels exactly as they do between any two fp32
                # implementations; everywhere else the difference is rounding noise)
                'rgb_difference_to_f32_kernels': {'max_abs': float((a_ - b_).abs().max()),
                                                  'rays_above_1e-4': int(((a_ - b_).abs().amax(-1) > 1e-4).sum()),
                                                  'median_abs': float((a_ - b_).abs().median())}}
        finally:
            rendering.SPLIT_PRECISION = False
        if args.mode == 'train':
            # opt-in split-precision TRAINING step: tape-writing forward + data-gradient chain on the 16-bit pipe (tapes, weight
            # gradients, heads, optimiser: the fp32 kernels).  Its own dtype; NOT `value`.
            from mega_nerf.training import FusedTrainStep
            fgm.train(), bgm.train()
            fs = FusedTrainStep([(fgm, bgm)], hp, sc, sr, args.rays, split_precision=True)
            t_ts = timed(lambda: fs([w['batch']]), args.steps, 3)
            fs.profile(8)
            for _ in range(8):
                fs([w['batch']])
            torch.cuda.synchronize()
            sp = [fs.kernel_times(i) for i in range(8)]
            # the split-precision weight-gradient launch is bound by the tape stream: algorithmic bytes = every dZ plane and every input
            # plane of every layer read once (floats per tape row: fg 4964, bg 5268 -- DESIGN.md section 3f)
            nbg_s = int(fs.n_bg[0])
            wg_bytes = 4.0 * (4964 * args.rays * (Nc + Nf) + 5268 * nbg_s * (Nc // 2 + Nf // 2))
            wg_ms = sum(d_['wgrad'] for d_ in sp) / len(sp)
            extras['train_split_precision'] = {
                'wgrad_roofline': {'bound': 'hbm', 'achieved': round(wg_bytes / wg_ms / 1e6, 1), 'peak': 8000.0, 'unit': 'GB/s',
                                   'frac': round(wg_bytes / wg_ms / 1e6 / 8000.0, 4), 'algorithmic_bytes_per_launch': int(wg_bytes),
                                   'kernel': 'k_wgrad2<0, true> (split-precision weight gradients of both models; one launch per step)',
                                   'avg_launch_ms': round(wg_ms, 4)},
                'dtype': 'forward, data-gradient chain and weight gradients: f16 hi/lo split operands, 3 f16 MFMA products per block, f32 accumulate '
                         '(gradients scaled by powers of two per row / per plane); tapes, heads, ray stages, Adam: f32 (opt-in; the f32 step is '
                         'the default and `value`)',
                'ms_per_step': t_ts * 1e3, 'rays_per_sec': args.rays / t_ts,
                'step_spans_ms': {k: round(sum(d_[k] for d_ in sp) / len(sp), 4) for k in sp[0]}}
            del fs
            fgm.eval(), bgm.eval()
        if args.mode == 'train' and not args.only_split_extras:
            # Launch quantisation, measured: the same fp32 step with the background branch of the forward on the plan's side stream, forked
            # behind the FOREGROUND's coarse pass (MNR_STEP_TWO_STREAMS=2, opt-in: csrc/step.hip).  The foreground's coarse launch -- 1024
            # workgroups = two whole rounds of the 512 resident slots -- then runs alone: its span is a clean single launch of the dominant
            # kernel WITHOUT the partial round the background's 69 workgroups add in the default one-stream schedule.
            from mega_nerf.training import FusedTrainStep
            fgm.train(), bgm.train()
            os.environ['MNR_STEP_TWO_STREAMS'] = '2'
            try:
                fs2 = FusedTrainStep([(fgm, bgm)], hp, sc, sr, args.rays)
            finally:
                os.environ.pop('MNR_STEP_TWO_STREAMS', None)
            t_2s = timed(lambda: fs2([w['batch']]), args.steps, 3)
            fs2.profile(8)
            for _ in range(8):
                fs2([w['batch']])
            torch.cuda.synchronize()
            sp2 = [fs2.kernel_times(i) for i in range(8)]
            fc_ms = sum(d_['fwd_c'] for d_ in sp2) / len(sp2)
            fc_fl = args.rays * Nc * FG_FLOP_PER_SAMPLE
            extras['train_background_branch_on_side_stream'] = {
                'ms_per_step': t_2s * 1e3, 'rays_per_sec': args.rays / t_2s,
                'foreground_coarse_launch_alone': {'kernel': 'k_mlp_fwd_multi<fg, bg, true>, %d fg rows = %d workgroups = %.2f rounds of 512 slots' % (
                    args.rays * Nc, args.rays * Nc // 64, args.rays * Nc / 64 / 512), 'avg_launch_ms': round(fc_ms, 4),
                    'algorithmic_gflop': round(fc_fl / 1e9, 2), 'achieved_tflops': round(fc_fl / fc_ms / 1e9, 2),
                    'frac_of_f32_mfma_peak': round(fc_fl / fc_ms / 1e9 / PEAK_F32_MFMA_TFLOPS, 4)},
                'step_spans_ms': {k: round(sum(d_[k] for d_ in sp2) / len(sp2), 4) for k in sp2[0]},
                'note': 'opt-in (MNR_STEP_TWO_STREAMS=2); `value` and `roofline` are the one-stream schedule, whose per-launch durations are not overlapped'}
            del fs2
            fgm.eval(), bgm.eval()
        if not args.only_split_extras:
            hp_ref = get_opts_base().parse_args([])                         # the reference's default 256 + 512 samples (opts.py:32-35)
            extras['eval_rays_per_sec_256+512_samples'] = args.rays / timed(ev_fn(w['batch'], hp_ref), 5, 1)
            fgm.train(), bgm.train()
            ts_ref = TrainStep(fgm, bgm, hp_ref, sc, sr)
            extras['train_rays_per_sec_256+512_samples'] = args.rays / timed(lambda: ts_ref(*w['batch']), 5, 4)     # (warm-up: 16 GB of tapes to allocate)
            del ts_ref
            # north-star PSNR check, GPU half (the CPU half runs inside cpu_baseline)
            prob = psnr_problem(hp, all_rays.cpu().numpy())
            psnr_here, tgt_train, tgt_test = psnr_gpu(hp, prob, dev)
            extras['_psnr_job'] = (prob, tgt_train, tgt_test)
            if 'train_split_precision' in extras:
                extras['train_split_precision']['psnr_student_vs_teacher_db'] = round(psnr_gpu(hp, prob, dev, split_step=True)[0], 4)
            extras['psnr'] = {'student_vs_teacher_db': round(psnr_here, 4),
                              'protocol': '%d Adam steps of %d rays (training mode: jitter + sigma noise, identical random numbers on both sides), '
                                          'PSNR of %d held-out rays against a fixed teacher field' % (PSNR_STEPS, PSNR_BATCH, PSNR_TEST_RAYS)}

    if rank == 0:
        total_rays = args.rays * args.steps * total_cells
        pmc = json.loads(PMC_FILE.read_text()) if PMC_FILE.exists() else {}
        n_fg_c, n_fg_f = args.rays * Nc, args.rays * Nf
        n_bg_c, n_bg_f = max(n_bg, 0) * (Nc // 2), max(n_bg, 0) * (Nf // 2)

        def roofline(tags, kernel, flops, pmc_key):
            """tags: the event tags of every launch of one kernel symbol in a step (``flops`` = mean per launch), so that
            avg_launch_ms is the same population as the kernel's row in a rocprofv3 trace."""
            ms = [a.elapsed_time(b) for t, a, b in ev if t in tags]
            for t in tags:                         # fused step: spans recorded by mnr_train_step itself (one launch covers all cells;
                ms += [v / (len(work) if t == 'wgrad' else 1) for v in span_ms.get(t, [])]      # the weight gradients: one per cell)
            if not ms:
                return None
            if span_ms and tags[0] != 'wgrad':
                flops = flops * len(work)
            avg = sum(ms) / len(ms) * 1e-3
            ach = flops / avg / 1e12
            p = pmc.get(pmc_key, {})
            return {'bound': 'mfma', 'achieved': round(ach, 2), 'peak': PEAK_F32_MFMA_TFLOPS, 'unit': 'TFLOP/s',
                    'frac': round(ach / PEAK_F32_MFMA_TFLOPS, 4), 'traffic': p.get('hbm_bytes_per_launch'),
                    'mfma_busy': p.get('mfma_busy'),
                    # traffic / mfma_busy are PMC figures from the committed summary (separate --pmc passes), NOT from this run:
                    'pmc_source': {'file': str(PMC_FILE.relative_to(ROOT)), 'git_head_when_summarised': pmc.get('_meta', {}).get('git_head_when_summarised')},
                    'kernel': kernel.split(' (')[0], 'kernel_detail': kernel, 'avg_launch_ms': round(avg * 1e3, 4),
                    'algorithmic_gflop_per_launch': round(flops / 1e9, 2)}

        mlp = lambda nf_, nb_: nf_ * FG_FLOP_PER_SAMPLE + nb_ * BG_FLOP_PER_SAMPLE                      # noqa: E731
        mfma_only = lambda nf_, nb_: mlp(nf_, nb_) - (nf_ + nb_) * HEAD_FLOP_PER_SAMPLE                # noqa: E731
        roof, extra_roof = None, {}
        if args.container:
            # whole-step figure from the device-side routed row counts (a row inside the boundary margin is evaluated by two cells)
            r_fg, r_bg = routed_timed
            mac_np = lambda w_: sum(v.size for k_, v in w_.items() if k_.endswith('weight') and not k_.startswith('embedding_a'))   # noqa: E731
            cn = work[0]['cells_np']                  # (FLOPs per sample from the cells' own shapes: 1 211 392 / 1 236 992 at W = 256)
            fl = (r_fg * 2.0 * mac_np(cn['fg'][0]) + r_bg * 2.0 * mac_np(cn['bg'][0])) / args.steps
            ach = fl / (dt / args.steps) / 1e12
            roof = {'bound': 'mfma', 'achieved': round(ach, 2), 'peak': PEAK_F32_MFMA_TFLOPS, 'unit': 'TFLOP/s',
                    'frac': round(ach / PEAK_F32_MFMA_TFLOPS, 4), 'traffic': None,
                    'kernel': 'whole step, wall clock',
                    'kernel_detail': 'whole step (k_mlp_fwd%s gather mode: all cells of a pass in one launch; k_route, k_route_combine, render stages), wall clock' % ('_pair' if args.layer_dim == 512 else ''),
                    'routed_rows_per_step': {'fg': r_fg // args.steps, 'bg': r_bg // args.steps,
                                             'unrouted': {'fg': args.rays * (Nc + Nf), 'bg': max(n_bg, 0) * (Nc // 2 + Nf // 2)}},
                    'algorithmic_gflop_per_step': round(fl / 1e9, 1)}
        elif wide:
            # whole-step figure: GEMM FLOPs of every MLP evaluation (x3 in training: forward, data and weight gradients) over
            # the step time -- the layer-by-layer path is ~40 launches per step, no single kernel dominates
            mac = lambda m_: sum(p.numel() for k_, p in m_.named_parameters() if k_.endswith('weight') and not k_.startswith('embedding_a'))   # noqa: E731
            fl = 2.0 * ((n_fg_c + n_fg_f) * mac(work[0]['fg']) + (n_bg_c + n_bg_f) * mac(work[0]['bg'])) * (3 if args.mode == 'train' else 1)
            ach = fl * len(work) / (dt / args.steps) / 1e12
            roof = {'bound': 'mfma', 'achieved': round(ach, 2), 'peak': PEAK_F32_MFMA_TFLOPS, 'unit': 'TFLOP/s',
                    'frac': round(ach / PEAK_F32_MFMA_TFLOPS, 4), 'traffic': None,
                    'kernel': 'whole step, wall clock',
                    'kernel_detail': ('whole step (k_tgemm forward / data-gradient launches + k_wgrad2<1>), wall clock incl. render stages' if args.mode == 'train' and args.sh_deg is None
                               else 'whole step (one-call step / render of the SH pair), wall clock incl. render stages' if args.sh_deg in (2, 3)
                               else 'whole step (register-chained kernels + stand-alone SH adjoint kernels, sequenced stage by stage), wall clock incl. render stages' if args.sh_deg is not None
                               else 'whole step (k_mlp_fwd_pair: 512-wide foreground, two wavefronts per SIMD; k_mlp_fwd background), wall clock incl. render stages'),
                    'algorithmic_gflop_per_step': round(fl * len(work) / 1e9, 1)}
        elif not args.container and (Nc, Nf) == (64, 128):
            if args.mode == 'train':
                # The three MFMA kernels of a training step.  `roofline` is the one with the LARGEST SHARE OF THE STEP TIME
                # (launches per step x average launch time); the other two follow under roofline_other_kernels.
                bwd_flops = mfma_only(n_fg_c + n_fg_f, n_bg_c + n_bg_f) - (n_fg_c + n_fg_f) * 2 * (75 + 5) * 256 \
                    - (n_bg_c + n_bg_f) * 2 * (100 + 5) * 256 - (n_fg_c + n_fg_f + n_bg_c + n_bg_f) * 2 * 27 * 128
                cands = [
                    (2, roofline(('fwd_c', 'fwd_f'), 'k_mlp_fwd_multi<fg, bg, true> (tape-writing forward: coarse + fine launch of a step, fg + bg '
                                 'rows; algorithmic FLOPs = mean of the two launches)', mlp(n_fg_c + n_fg_f, n_bg_c + n_bg_f) / 2, 'k_mlp_fwd_multi_train')),
                    (1, roofline(('wgrad',), 'k_wgrad2 (fg %d + bg %d rows, weight gradients of every layer of both models; one launch per step)' % (
                        n_fg_c + n_fg_f, n_bg_c + n_bg_f), mfma_only(n_fg_c + n_fg_f, n_bg_c + n_bg_f), 'k_wgrad2')),
                    (1, roofline(('bwd',), 'k_mlp_bwd_multi<fg, bg> (data-gradient chains of all four segments of a step; one launch)', bwd_flops,
                                 'k_mlp_bwd_multi'))]
                cands = [(n_, r_) for n_, r_ in cands if r_ is not None]
                for n_, r_ in cands:
                    nl_ = n_ * (len(work) if (not span_ms or r_['kernel'].startswith('k_wgrad2')) else 1)
                    r_['launches_per_step'] = nl_
                    r_['share_of_step_time'] = round(nl_ * r_['avg_launch_ms'] / (dt / args.steps * 1e3), 4)
                cands.sort(key=lambda c_: -c_[1]['share_of_step_time'])
                if cands:
                    roof = cands[0][1]
                    extra_roof = {c_[1]['kernel'].split(' ')[0]: c_[1] for c_ in cands[1:]}
                    fl_step = 3 * mlp(n_fg_c + n_fg_f, n_bg_c + n_bg_f) * len(work)
                    extras['whole_step_vs_mfma_ideal'] = {
                        'algorithmic_gflop_per_step': round(fl_step / 1e9, 1), 'tflops': round(fl_step / (dt / args.steps) / 1e12, 1),
                        'frac_of_f32_mfma_peak': round(fl_step / (dt / args.steps) / 1e12 / PEAK_F32_MFMA_TFLOPS, 4)}
            else:
                roof = roofline(('fwd_coarse', 'fwd_fine'), 'k_mlp_fwd_multi<fg, bg, false> (coarse + fine launch of a step, fg + bg rows)',
                                mlp(n_fg_c + n_fg_f, n_bg_c + n_bg_f) / 2, 'k_mlp_fwd_multi_eval')
                extra_roof = {'fine launch only': roofline(('fwd_fine',), 'k_mlp_fwd_multi<fg, bg, false>', mlp(n_fg_f, n_bg_f),
                                                           'k_mlp_fwd_multi_eval_fine')}
        cpu = None
        if not args.no_cpu_baseline and world == 1 and args.container and not wide:
            cpu = cpu_baseline_container(hp, work[0]['batch'][0].cpu().numpy(), work[0]['batch'][1].cpu().numpy(), work[0]['cells_np'])
        if not args.no_cpu_baseline and world == 1 and not args.container and not wide:          # rank 0 at N = 1 only
            w = work[0]
            b = w['batch']
            cpu = cpu_baseline(hp, b[0].cpu().numpy(), b[1].cpu().numpy(), b[2].cpu().numpy(), w['fw'], w['bw'], w['fcfg'], w['bcfg'],
                               min(1024, args.rays), args.mode, extras.get('_psnr_job'))
            if 'psnr' in extras and 'psnr_db' in cpu:
                extras['psnr']['cpu_restatement_db'] = round(cpu.pop('psnr_db'), 4)
                extras['psnr']['abs_difference_db'] = round(abs(extras['psnr']['student_vs_teacher_db'] - extras['psnr']['cpu_restatement_db']), 4)
                if 'psnr_student_vs_teacher_db' in extras.get('train_split_precision', {}):
                    extras['train_split_precision']['psnr_abs_difference_to_cpu_restatement_db'] = round(
                        abs(extras['train_split_precision']['psnr_student_vs_teacher_db'] - extras['psnr']['cpu_restatement_db']), 4)
                extras['psnr']['cpu_seconds'] = cpu.pop('psnr_seconds')
        extras.pop('_psnr_job', None)
        if args.container:
            shard = 'merged %d-cell container (MegaNeRF router, margin 1.15) routed on one GPU' % args.container
        elif args.submodules:
            shard = '%d submodules dealt round-robin to %d GPU(s); every GPU steps through its cells' % (args.submodules, world)
        else:
            shard = 'one submodule per GPU'
        line = {
            'metric': 'train rays/sec (fwd+bwd+2xAdam step)' if args.mode == 'train' else 'eval rays/sec (render_rays fwd)',
            'value': total_rays / dt, 'unit': 'rays/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': dt / args.steps * 1e3, 'higher_is_better': True, 'scaling': 'strong' if args.submodules else 'weak',
            'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': '%s cell, fg 8x%d + bg 8x%d, %d rays x (%d+%d) samples, %s' % (
                           'configs[4] Sci-Art-shaped sh_deg %d' % args.sh_deg if args.sh_deg is not None else 'configs[3] Building-shaped' if wide
                           else 'configs[1] Rubble-shaped' if not (args.submodules or args.container) else 'configs[2] Rubble-shaped',
                           args.layer_dim, hp.bg_layer_dim, args.rays, Nc, Nf,
                           '%d-cell container' % args.container if args.container else '%d submodules on %d GPU(s)' % (args.submodules, world) if args.submodules
                           else 'one submodule per GPU'),
                       'workload_detail': ('configs/mega-nerf-sh-3 Sci-Art-shaped fg+bg NeRF (sh_deg %d, pos_dir_dim 0, fg 8x%d, bg 8x%d, 12 freqs, 48-d appearance), ' % (
                                        args.sh_deg, args.layer_dim, hp.bg_layer_dim) if args.sh_deg is not None else
                                    'configs/mega-nerf %s-shaped fg+bg NeRF (fg 8x%d, bg 8x%d, 12/4 freqs, 48-d appearance), ' % (
                                        'Building' if wide else 'Rubble', args.layer_dim, hp.bg_layer_dim)) +
                                   '%d rays x (%d+%d) samples per submodule step, %s' % (args.rays, Nc, Nf, shard),
                       'mode': args.mode, 'rays_per_batch': args.rays, 'bg_rays_in_batch': n_bg, 'submodules': total_cells,
                       'parallelism': 'submodule-per-gpu x%d' % world if not args.submodules else 'submodules %d over %d gpus' % (args.submodules, world)},
            'metric_allreduce_check_db': round(metric_reduce_check, 4), 'host': host_diag,
            'roofline': roof, 'cpu_baseline': cpu,
        }
        if extra_roof and any(v is not None for v in extra_roof.values()):
            line['roofline_other_kernels'] = extra_roof
        if span_ms:
            line['step_spans_ms'] = {k: round(sum(v) / len(v), 4) for k, v in span_ms.items()}
            mlp_ms = sum(line['step_spans_ms'][k] for k in ('fwd_c', 'fwd_f', 'bwd', 'wgrad'))
            line['step_spans_ms']['non_mlp_share_of_step'] = round(1.0 - mlp_ms / (dt / args.steps * 1e3), 4)
            line['host']['launches_per_step'] = 11 + 2 * len(work)      # memset + 10 kernels + (k_wgrad2 + reduce) per cell
        # self-diagnosis: how the timed regions spread, what the box delivered right before / after them, what the clocks did meanwhile
        diag = {'timing': timing}
        if cal_before is not None:
            diag['calibration'] = {'before': cal_before, 'after': cal_after, 'device': dev_info,
                                   'nominal': {'mfma_f32_tflops': PEAK_F32_MFMA_TFLOPS, 'note': 'csrc/calibrate.hip; typical MI355X in this pool: mfma_f32_tflops ~145 '
                                               '(1 ms probe incl. clock ramp), sclk_mhz_mfma_chain 2400, dma_chunk_round_trip_us ~0.49, chase_l2 / mall / hbm ns, hbm GB/s: see '
                                               'profiles/r06_calibration_healthy_box.jsonl'}}
        if cal_before is not None and 'error' not in cal_before:
            # a one-line reading of the calibration against what healthy boxes of this pool deliver (profiles/r06_calibration_healthy_box.jsonl)
            flags = []
            for when, c in (('before', cal_before), ('after', cal_after or {})):
                if not c or 'error' in c:
                    continue
                if c.get('mfma_f32_tflops', 1e9) < 135:
                    flags.append('%s: fp32-MFMA probe %.0f TFLOP/s (healthy 142-147)' % (when, c['mfma_f32_tflops']))
                if c.get('mfma_wg_ms_median') and c['mfma_wg_ms_max'] > 1.10 * c['mfma_wg_ms_median']:
                    flags.append('%s: slowest workgroup %.2fx the median (healthy <= 1.03): a CU / XCD runs behind' % (when, c['mfma_wg_ms_max'] / c['mfma_wg_ms_median']))
                if c.get('mfma_xcd_ms_fastest') and c['mfma_xcd_ms_slowest'] > 1.06 * c['mfma_xcd_ms_fastest']:
                    flags.append('%s: slowest XCD %.2fx the fastest (healthy <= 1.03)' % (when, c['mfma_xcd_ms_slowest'] / c['mfma_xcd_ms_fastest']))
                if c.get('sclk_mhz_mfma_chain', 1e9) < 2250:
                    flags.append('%s: one wavefront sees %.0f MHz (healthy 2395-2415)' % (when, c['sclk_mhz_mfma_chain']))
                if c.get('dma_chunk_round_trip_us', 0) > 0.7:
                    flags.append('%s: 32 KiB L2 -> LDS chunk round trip %.2f us (healthy 0.48-0.49)' % (when, c['dma_chunk_round_trip_us']))
                if c.get('chase_l2_ns', 0) > 280 or c.get('chase_hbm_ns', 0) > 450:
                    flags.append('%s: load latency L2 %.0f / HBM %.0f ns (healthy 218 / 335-350)' % (when, c.get('chase_l2_ns', 0), c.get('chase_hbm_ns', 0)))
                if c.get('hbm_read_gbps', 1e9) < 6000:
                    flags.append('%s: HBM stream read %.0f GB/s (healthy 6900-7050)' % (when, c['hbm_read_gbps']))
            diag['box_verdict'] = 'calibration within the healthy range of this pool' if not flags else 'BOX BELOW NOMINAL -- ' + '; '.join(flags)
        cl = clocks.summary(windows)
        if cl is not None:
            diag['clocks_during_timed_regions'] = cl
        if xcd is not None:
            diag['xcd_clocks_under_load'] = xcd
        diag['ranks'] = rank_info
        # the runtime-relevant environment of this process (a setting of the launching shell that changes kernel behaviour shows here)
        diag['env'] = {k: v for k, v in sorted(os.environ.items()) if k.startswith(('HIP_', 'HSA_', 'ROCR_', 'GPU_', 'AMD_', 'ROC_', 'MNR_', 'NCCL_', 'RCCL_', 'PYTORCH_', 'OMP_NUM'))}
        line['diag'] = diag
        if roof is not None:
            # (the driver's record keeps `roofline` whole and only the tail of the rest of the line: the figures needed to tell a slow box
            # from slow code ride inside it)
            roof['timing'] = {k: timing[k] for k in ('regions_ms_per_step', 'min', 'median', 'max') if k in timing}
            if 'fwd_launch_ms_by_region' in timing:
                roof['timing']['fwd_launch_ms_by_region'] = timing['fwd_launch_ms_by_region']
            if cal_before is not None:
                keys = ('mfma_f32_tflops', 'sclk_mhz_mfma_chain', 'dma_stream_gbps', 'dma_chunk_round_trip_us', 'chase_l2_ns', 'chase_mall_ns', 'chase_hbm_ns',
                        'hbm_read_gbps', 'hbm_write_gbps')
                roof['box'] = {'before': {k: cal_before.get(k) for k in keys}, 'after': {k: (cal_after or {}).get(k) for k in keys},
                               'verdict': diag.get('box_verdict')}
                # the same achieved rate against what THIS box delivered on a pure fp32-MFMA loop chip-wide (clocks under that load
                # are power-limited below the 2.4 GHz the 157.3 TFLOP/s peak is priced at): context, not a second roofline
                rates = [c.get('mfma_f32_tflops') for c in (cal_before, cal_after or {}) if c.get('mfma_f32_tflops')]
                if rates and roof.get('achieved'):
                    roof['box']['achieved_over_calibrated_mfma_rate'] = round(roof['achieved'] / (sum(rates) / len(rates)), 4)
                if cl is not None:
                    roof['box']['sclk_mhz_during'] = next((v for k, v in cl.items() if k.startswith('sclk')), None)
                    roof['box']['power_w_during'] = cl.get('power_w')
            if span_ms:
                roof['step_spans_ms'] = line['step_spans_ms']
        line.update(extras)
        line['diag'] = line.pop('diag')          # last key of the line: the driver's record keeps the tail
        return line
    return None


if __name__ == '__main__':
    main()
