"""End-to-end drop-in check on the GPU: synthetic dataset in the reference's on-disk layout -> train.main ->
checkpoint with the reference's keys -> eval.main -> metrics.txt."""
import os
import re
import subprocess
import sys
from pathlib import Path

import pytest
from conftest import free_port, loopback_env
import torch

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


def test_train_then_eval_entry_points(tmp_path):
    from mega_nerf import eval as ev
    from mega_nerf import train as tr
    from mega_nerf.opts import get_opts_base
    data = tmp_path / 'data'
    subprocess.run([sys.executable, str(ROOT / 'mega-nerf_amd' / 'tools' / 'make_synthetic_dataset.py'), '--out', str(data),
                    '--images', '8', '--val_every', '4', '--size', '32', '--samples', '32', '64'], check=True)
    assert (data / 'coordinates.pt').exists() and len(list((data / 'val' / 'metadata').iterdir())) == 2
    common = ['--dataset_path', str(data), '--coarse_samples', '32', '--fine_samples', '64', '--near', '0.01',
              '--ray_altitude_range', '-0.5', '0.2', '--val_scale_factor', '1', '--batch_size', '512']

    def parse(extra):
        p = get_opts_base()
        p.add_argument('--exp_name', type=str, required=True)
        p.add_argument('--dataset_path', type=str, required=True)
        return p.parse_args(common + extra)

    exp = tmp_path / 'exp'
    tr.main(parse(['--exp_name', str(exp), '--train_iterations', '40', '--ckpt_interval', '20']))
    run0 = exp / '0'
    ck = torch.load(run0 / 'models' / '40.pt', map_location='cpu', weights_only=False)
    for key in ('model_state_dict', 'bg_model_state_dict', 'optimizers', 'iteration', 'torch_random_state',
                'np_random_state', 'random_state', 'dataset_index', 'scaler'):
        assert key in ck, key
    assert ck['iteration'] == 40 and 'xyz_encodings.0.0.weight' in ck['model_state_dict']
    assert (run0 / 'models' / '20.pt').exists()
    m_train = (run0 / 'metrics.txt').read_text()
    assert 'Average val/psnr' in m_train
    for f in ('hparams.txt', 'command.txt', 'image_indices.txt'):
        assert (run0 / f).exists()
    # the validation panels the reference hands to TensorBoard (runner.py:452-491): ground truth | render | depth, + bg / fg panels
    from PIL import Image
    for name in ('0.jpg', '1.jpg', '0_bg.jpg', '0_fg.jpg'):
        assert Image.open(run0 / 'val_images' / '40' / name).size == (3 * 32, 32), name
    # evaluation of the checkpoint through eval.py reproduces the validation PSNR written after training
    ev.main(parse(['--exp_name', str(exp), '--ckpt_path', str(run0 / 'models' / '40.pt')]))
    m_eval = (exp / '1' / 'metrics.txt').read_text()
    def metric(text, key):
        return float([ln for ln in text.splitlines() if ln.startswith('Average ' + key)][0].split(':')[1])

    a, b = metric(m_train, 'val/psnr'), metric(m_eval, 'val/psnr')
    assert abs(a - b) < 0.05, (a, b)            # north star: PSNR within 0.05 dB
    assert a > 5.0
    assert abs(metric(m_train, 'val/ssim') - metric(m_eval, 'val/ssim')) < 1e-3 and 0.0 < metric(m_eval, 'val/ssim') <= 1.0


def _dataset(tmp_path):
    data = tmp_path / 'data'
    subprocess.run([sys.executable, str(ROOT / 'mega-nerf_amd' / 'tools' / 'make_synthetic_dataset.py'), '--out', str(data),
                    '--images', '8', '--val_every', '4', '--size', '32', '--samples', '64', '128'], check=True)
    return data


def _hparams(data, exp, extra):
    from mega_nerf.opts import get_opts_base
    p = get_opts_base()
    p.add_argument('--exp_name', type=str, required=True)
    p.add_argument('--dataset_path', type=str, required=True)
    return p.parse_args(['--dataset_path', str(data), '--exp_name', str(exp), '--coarse_samples', '64', '--fine_samples', '128', '--near', '0.01',
                         '--ray_altitude_range', '-0.5', '0.2', '--val_scale_factor', '1', '--batch_size', '384'] + extra)


@pytest.mark.parametrize('width', [256, 512])
def test_runner_train_runs_the_fused_step_and_resumes(tmp_path, monkeypatch, width):
    """(``width`` 512: the Building shape -- the same through the 512-wide foreground's path inside mnr_train_step.)
    train.py's loop (Runner.train, reference runner.py:244-277) on the default architecture must run the ONE-CALL training step
    (mnr_train_step through training.CellTrainer) -- the thing bench.py times -- and stay a drop-in:
      * same trained weights as the stage-by-stage autograd loop (MNR_RUNNER_AUTOGRAD=1) from the same seed, on deterministic
        renders (models pinned to eval mode: the two paths draw their random numbers from different generators);
      * 7168 training pixels in batches of 384 leave a ragged last batch of 256 rays per epoch: those steps take the autograd path
        INSIDE the fused trainer, on the same Adam moments / step counts / learning rate;
      * the checkpoint keeps the reference's keys and its `optimizers` entry loads into a plain torch.optim.Adam;
      * a run resumed from its own 20-iteration checkpoint ends where the uninterrupted run ends."""
    import numpy as np
    from mega_nerf.models.nerf import NeRF
    from mega_nerf.runner import Runner
    data = _dataset(tmp_path)
    monkeypatch.setattr(NeRF, 'train', lambda self, mode=True: torch.nn.Module.train(self, False))

    def run(tag, extra, autograd=False):
        if autograd:
            monkeypatch.setenv('MNR_RUNNER_AUTOGRAD', '1')
        else:
            monkeypatch.delenv('MNR_RUNNER_AUTOGRAD', raising=False)
        r = Runner(_hparams(data, tmp_path / tag, ['--train_iterations', '40', '--ckpt_interval', '20', '--layer_dim', str(width)] + extra))
        w0 = {k: v.detach().clone() for k, v in list(r.nerf.state_dict().items()) + [('bg.' + k, v) for k, v in r.bg_nerf.state_dict().items()]}
        r.train()
        w = {k: v.detach().clone() for k, v in list(r.nerf.state_dict().items()) + [('bg.' + k, v) for k, v in r.bg_nerf.state_dict().items()]}
        return r, w0, w

    ra, w0, wa = run('fused', [])
    assert ra.trainer is not None and ra.trainer.fused is not None and ra.trainer.fused.n_rays == 384
    assert ra.trainer.iteration == 40
    rb, w0b, wb = run('autograd', [], autograd=True)
    assert rb.trainer is None
    for k in w0:
        np.testing.assert_array_equal(w0[k].cpu().numpy(), w0b[k].cpu().numpy())
    # 40 Adam steps amplify rounding differences between two implementations of the same gradients (|update| ~ lr whatever the
    # gradient's size): compare the MOVEMENT of every tensor
    worst = {}
    for k in wa:
        moved = float((wb[k] - w0[k]).norm())
        if moved > 0:
            worst[k] = float((wa[k] - wb[k]).norm()) / moved
    print('fused vs autograd loop, |dw| / |movement|:', {k: '%.2e' % v for k, v in sorted(worst.items(), key=lambda kv: -kv[1])[:5]})
    assert max(worst.values()) < 0.05, worst
    ck = torch.load(tmp_path / 'fused' / '0' / 'models' / '40.pt', map_location='cpu', weights_only=False)
    for key in ('model_state_dict', 'bg_model_state_dict', 'optimizers', 'iteration', 'torch_random_state', 'np_random_state', 'random_state',
                'dataset_index', 'scaler'):
        assert key in ck, key
    ckb = torch.load(tmp_path / 'autograd' / '0' / 'models' / '40.pt', map_location='cpu', weights_only=False)
    for key in ('nerf', 'bg_nerf'):
        sa, sb = ck['optimizers'][key], ckb['optimizers'][key]
        assert sa['param_groups'][0].keys() == sb['param_groups'][0].keys() and sa['state'].keys() == sb['state'].keys()
        assert abs(sa['param_groups'][0]['lr'] - sb['param_groups'][0]['lr']) < 1e-12
        for i in sa['state']:
            assert float(sa['state'][i]['step']) == float(sb['state'][i]['step']) == 40.0
            assert sa['state'][i]['exp_avg'].shape == sb['state'][i]['exp_avg'].shape
        # ... and the entry loads into a plain torch.optim.Adam over a fresh model, the way the reference resumes (runner.py:181-184)
        m = (ra.nerf if key == 'nerf' else ra.bg_nerf)
        opt = torch.optim.Adam(m.parameters(), lr=5e-4)
        sd = opt.state_dict()
        sd.update(sa)
        opt.load_state_dict(sd)
        assert float(opt.state[next(iter(m.parameters()))]['step']) == 40.0
    # resume from the run's own 20-iteration checkpoint: same weights at iteration 40 as the uninterrupted run (kernel sums are
    # not bit-reproducible run to run: atomics in the head / embedding gradients)
    rc, _, wc = run('resumed', ['--ckpt_path', str(tmp_path / 'fused' / '0' / 'models' / '20.pt')])
    assert rc.trainer.fused is not None and rc.trainer.iteration == 40
    worst = {k: float((wa[k] - wc[k]).norm()) / max(float((wa[k] - w0[k]).norm()), 1e-30) for k in wa if float((wa[k] - w0[k]).norm()) > 0}
    print('resumed vs uninterrupted:', {k: '%.2e' % v for k, v in sorted(worst.items(), key=lambda kv: -kv[1])[:5]})
    assert max(worst.values()) < 0.02, worst


def test_sh_config_trains_through_the_fused_step_and_evaluates(tmp_path):
    """configs/mega-nerf-sh-3 (sh_deg 2, pos_dir_dim 0) end to end through the reference's entry points: train.py runs the one-call step
    (k_sh_head_bwd inside mnr_train_step), the checkpoint evaluates through eval.py (mnr_render_fwd with the SH epilogue) to the PSNR
    written after training."""
    from mega_nerf import eval as ev
    from mega_nerf.runner import Runner
    data = _dataset(tmp_path)
    sh = ['--sh_deg', '2', '--pos_dir_dim', '0', '--batch_size', '512']
    r = Runner(_hparams(data, tmp_path / 'exp', ['--train_iterations', '30', '--ckpt_interval', '30'] + sh))
    r.train()
    assert r.trainer is not None and r.trainer.fused is not None and r.nerf.rgb_dim == 27
    run0 = tmp_path / 'exp' / '0'
    m_train = (run0 / 'metrics.txt').read_text()
    ev.main(_hparams(data, tmp_path / 'exp', ['--ckpt_path', str(run0 / 'models' / '30.pt')] + sh))
    m_eval = (tmp_path / 'exp' / '1' / 'metrics.txt').read_text()

    def metric(text, key):
        return float([ln for ln in text.splitlines() if ln.startswith('Average ' + key)][0].split(':')[1])
    a, b = metric(m_train, 'val/psnr'), metric(m_eval, 'val/psnr')
    assert abs(a - b) < 0.05 and a > 5.0, (a, b)


def test_render_images_script_writes_the_reference_tree(tmp_path):
    """scripts/render_images.py (reference scripts/render_images.py:19-144): poses / intrinsics / embeddings text files in, rgbs / depths /
    cells (+ depths_npz) out; the rgb files are the Runner.render_image colours of the same pose (up to JPEG), --resume skips finished poses."""
    import importlib.util
    import numpy as np
    from PIL import Image
    from mega_nerf import train as tr
    from mega_nerf.image_metadata import ImageMetadata
    from mega_nerf.runner import Runner
    data = _dataset(tmp_path)
    exp = tmp_path / 'exp'
    tr.main(_hparams(data, exp, ['--train_iterations', '10', '--ckpt_interval', '10']))
    ckpt = exp / '0' / 'models' / '10.pt'
    inp = tmp_path / 'poses'
    inp.mkdir()
    mds = [torch.load(p, map_location='cpu', weights_only=False) for p in sorted((data / 'val' / 'metadata').iterdir())]
    (inp / 'poses.txt').write_text('\n'.join(' '.join('%.9g' % float(x) for x in md['c2w'].reshape(-1)) for md in mds) + '\n')
    (inp / 'intrinsics.txt').write_text('\n'.join('%d %d ' % (md['W'], md['H']) + ' '.join('%.9g' % float(x) for x in md['intrinsics']) for md in mds) + '\n')
    (inp / 'embeddings.txt').write_text('\n'.join(str(3 * k) for k in range(len(mds))) + '\n')
    cen = tmp_path / 'params.pt'
    torch.save({'centroids': torch.tensor([[0., -0.3, -0.3], [0., 0.3, -0.3], [0., -0.3, 0.3], [0., 0.3, 0.3]])}, cen)
    spec = importlib.util.spec_from_file_location('render_images', ROOT / 'mega-nerf_amd' / 'scripts' / 'render_images.py')
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    out = tmp_path / 'renders'
    argv = ['--dataset_path', str(data), '--coarse_samples', '64', '--fine_samples', '128', '--near', '0.01', '--ray_altitude_range', '-0.5', '0.2',
            '--val_scale_factor', '1', '--ckpt_path', str(ckpt), '--input', str(inp), '--output', str(out), '--centroids_path', str(cen),
            '--save_depth_npz']
    mod.main(mod._get_render_opts(argv))
    n = len(mds)
    for sub, ext in (('rgbs', 'jpg'), ('depths', 'jpg'), ('cells', 'jpg'), ('depths_npz', 'npy')):
        assert sorted(p.name for p in (out / sub).iterdir()) == ['%06d.%s' % (i, ext) for i in range(n)], sub
    # the written colours are the renderer's
    hp = mod._get_render_opts(argv)
    r = Runner(hp, False)
    r.nerf.eval(), r.bg_nerf.eval()
    md = mds[1]
    with torch.inference_mode():
        res, _ = r.render_image(ImageMetadata(Path(''), md['c2w'], md['W'], md['H'], md['intrinsics'], 3, None, False))
    want = (res['rgb_fine'].view(md['H'], md['W'], 3) * 255).byte().cpu().numpy().astype(np.float64)
    got = np.array(Image.open(out / 'rgbs' / '000001.jpg')).astype(np.float64)
    assert got.shape == want.shape
    assert 10 * np.log10(255.0 ** 2 / np.mean((got - want) ** 2)) > 28.0           # JPEG quality 75 of a smooth 32 x 32 render
    depth = np.load(out / 'depths_npz' / '000001.npy')
    np.testing.assert_allclose(depth, torch.nan_to_num(res['depth_fine']).view(md['H'], md['W']).cpu().numpy() * r.pose_scale_factor, rtol=1e-5)
    vis = np.array(Image.open(out / 'cells' / '000000.jpg'))
    assert vis.shape == (md['H'], md['W'], 3)
    # --resume: finished poses are skipped (their files keep their timestamps), a missing one is rendered again
    stamp = (out / 'rgbs' / '000000.jpg').stat().st_mtime_ns
    (out / 'cells' / '000001.jpg').unlink()
    mod.main(mod._get_render_opts(argv + ['--resume']))
    assert (out / 'rgbs' / '000000.jpg').stat().st_mtime_ns == stamp and (out / 'cells' / '000001.jpg').exists()
    # without --resume an existing output tree is refused, as the reference's mkdir(exist_ok=False) does
    with pytest.raises(FileExistsError):
        mod.main(mod._get_render_opts(argv))


def test_train_cells_job_two_ranks_merges_in_job_and_evaluates(tmp_path):
    """The north-star multi-GPU job (`mega-nerf_amd/tools/train_cells.py`; reference: parscripts/run_8.txt:1-8 + scripts/merge_submodules.py:33-78
    + runner.py:495-510) launched by `torch.distributed.run` with 2 ranks -- sharing this box's one GPU over gloo, the way
    test_two_ranks_step_the_fixed_eight_cell_set does (RCCL refuses two ranks on one device): 4 cells of a 2 x 2 grid dealt 2 + 2, each
    trained on its cluster-masked pixels with no collective, merged by ONE all_gather, evaluated image-parallel with one all_reduce.
      * the in-job container is BIT-IDENTICAL to the one scripts/merge_submodules.py builds from the same ranks' checkpoint files;
      * every rank ends with the same all-reduced validation PSNR / SSIM, and it is what a single-process eval of the container gives."""
    import json
    import os
    data = tmp_path / 'data'
    tools = ROOT / 'mega-nerf_amd' / 'tools'
    scripts = ROOT / 'mega-nerf_amd' / 'scripts'
    subprocess.run([sys.executable, str(tools / 'make_synthetic_dataset.py'), '--out', str(data), '--images', '8', '--val_every', '4',
                    '--size', '32', '--samples', '32', '64'], check=True)
    flags = ['--dataset_path', str(data), '--coarse_samples', '64', '--fine_samples', '128', '--near', '0.01', '--ray_altitude_range', '-0.5', '0.2',
             '--val_scale_factor', '1', '--boundary_margin', '1.5']
    masks = tmp_path / 'masks'
    subprocess.run([sys.executable, str(scripts / 'create_cluster_masks.py'), '--output', str(masks), '--grid_dim', '2', '2', '--ray_samples', '64'] + flags,
                   check=True)
    assert (masks / 'params.pt').exists() and sorted(p.name for p in masks.iterdir() if p.is_dir()) == ['0', '1', '2', '3']
    exp = tmp_path / 'job'
    env = loopback_env(dict(os.environ, MNR_SHARE_GPU='1', MASTER_ADDR='127.0.0.1', HSA_ENABLE_IPC_MODE_LEGACY='0', MNR_NO_VAL_IMAGES='1'))
    train_flags = flags + ['--train_iterations', '6', '--ckpt_interval', '6', '--val_interval', '1000', '--batch_size', '256']
    def launch():
        return subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
                               '--master-port', free_port(), str(tools / 'train_cells.py'), '--mask_path', str(masks), '--exp_name', str(exp)] + train_flags,
                              env=env, capture_output=True, text=True, timeout=900)
    r = launch()
    if r.returncode != 0 and any(t in r.stderr for t in ('EADDRINUSE', 'address already in use', 'RendezvousConnectionError', 'Connection refused')):
        # (the launcher's rendezvous, not the job: one more attempt on a new port, loudly)
        import shutil
        import warnings
        warnings.warn('train_cells.py: rendezvous failed, launching again: ' + r.stderr[-1500:])
        for d in tmp_path.glob('job*'):
            shutil.rmtree(d) if d.is_dir() else d.unlink()
        r = launch()
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    lines = [json.loads(ln.split('TRAIN_CELLS ', 1)[1]) for ln in r.stdout.splitlines() if 'TRAIN_CELLS ' in ln]
    assert sorted(ln['rank'] for ln in lines) == [0, 1] and {tuple(ln['cells']) for ln in lines} == {(0, 2), (1, 3)}
    assert lines[0]['val_psnr'] == lines[1]['val_psnr'] and lines[0]['val_ssim'] == lines[1]['val_ssim'] and lines[0]['val_psnr'] > 3.0
    for j in range(4):                                   # checkpoints where the reference's per-cell runs put them
        assert (tmp_path / 'job-{}'.format(j) / '0' / 'models' / '6.pt').exists(), j
    assert 'Average val/psnr' in (tmp_path / 'job-eval' / '0' / 'metrics.txt').read_text()
    # the reference's file hand-off over the same checkpoints
    by_file = tmp_path / 'by_file.pt'
    subprocess.run([sys.executable, str(scripts / 'merge_submodules.py'), '--ckpt_prefix', str(tmp_path / 'job-'), '--centroid_path', str(masks / 'params.pt'),
                    '--output', str(by_file), '--train_iterations', '6'] + flags[2:], check=True, env=env)
    a = torch.jit.load(str(tmp_path / 'job-merged.pt'), map_location='cpu').state_dict()
    b = torch.jit.load(str(by_file), map_location='cpu').state_dict()
    assert a.keys() == b.keys() and len(a) > 100
    for k in a:
        assert torch.equal(a[k], b[k]), k
    # a single-process evaluation of the merged container reproduces the job's all-reduced metrics
    from mega_nerf import eval as ev
    from mega_nerf.opts import get_opts_base
    p = get_opts_base()
    p.add_argument('--exp_name', type=str, required=True)
    p.add_argument('--dataset_path', type=str, required=True)
    ev.main(p.parse_args(flags + ['--exp_name', str(tmp_path / 'solo'), '--container_path', str(tmp_path / 'job-merged.pt')]))
    solo = (tmp_path / 'solo' / '0' / 'metrics.txt').read_text()
    solo_psnr = float([ln for ln in solo.splitlines() if ln.startswith('Average val/psnr')][0].split(':')[1])
    assert abs(solo_psnr - lines[0]['val_psnr']) < 1e-4, (solo_psnr, lines[0]['val_psnr'])


def test_a_ranks_cells_side_by_side_in_one_plan_train_like_one_after_the_other(tmp_path):
    """tools/train_cells.py on ONE rank that owns all four cells of a 2 x 2 grid (Building: 25 cells on 8 GPUs = 4,3,3,...): the default --
    the cells' unchanged Runner.train() loops on host threads, their iterations meeting in ONE `mnr_train_step` call per iteration
    (training.JointCells), each cell on its own cluster-masked dataset with its own random streams -- against `--sequential_cells` (one
    cell after the other, the reference's parscripts/run_8.txt layout on one GPU).  Same batches, same random numbers: the merged
    containers agree to the summation order of the gradients' partial sums (a weight moves by at most lr per Adam step; two orders of a
    noise-level gradient may disagree on its sign), and nearly all weights agree far better than that.  The joint job must actually have
    taken joint steps, and its ragged last batches of an epoch the cell-by-cell path."""
    data = tmp_path / 'data'
    tools = ROOT / 'mega-nerf_amd' / 'tools'
    scripts = ROOT / 'mega-nerf_amd' / 'scripts'
    subprocess.run([sys.executable, str(tools / 'make_synthetic_dataset.py'), '--out', str(data), '--images', '8', '--val_every', '4',
                    '--size', '32', '--samples', '32', '64'], check=True)
    flags = ['--dataset_path', str(data), '--coarse_samples', '64', '--fine_samples', '128', '--near', '0.01', '--ray_altitude_range', '-0.5', '0.2',
             '--val_scale_factor', '1', '--boundary_margin', '1.5']
    masks = tmp_path / 'masks'
    subprocess.run([sys.executable, str(scripts / 'create_cluster_masks.py'), '--output', str(masks), '--grid_dim', '2', '2', '--ray_samples', '64'] + flags,
                   check=True)
    iters = 8
    env = dict(os.environ, MNR_NO_VAL_IMAGES='1')
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK'):
        env.pop(k, None)
    train_flags = flags + ['--train_iterations', str(iters), '--ckpt_interval', str(iters), '--val_interval', '1000', '--batch_size', '256', '--skip_eval']
    outs = {}
    for mode, extra in (('joint', []), ('seq', ['--sequential_cells'])):
        r = subprocess.run([sys.executable, str(tools / 'train_cells.py'), '--mask_path', str(masks), '--exp_name', str(tmp_path / mode)] + train_flags + extra,
                           env=env, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
        outs[mode] = r.stdout
        for j in range(4):
            assert (tmp_path / '{}-{}'.format(mode, j) / '0' / 'models' / '{}.pt'.format(iters)).exists(), (mode, j)
    line = [ln for ln in outs['joint'].splitlines() if 'side by side' in ln]
    assert len(line) == 1, outs['joint'][-2000:]
    joint_steps, separate = [int(v) for v in re.findall(r'(\d+) joint steps, (\d+) cell-by-cell', line[0])[0]]
    assert joint_steps + separate == iters and joint_steps >= iters - 2, line[0]
    a = torch.jit.load(str(tmp_path / 'joint-merged.pt'), map_location='cpu').state_dict()
    b = torch.jit.load(str(tmp_path / 'seq-merged.pt'), map_location='cpu').state_dict()
    assert a.keys() == b.keys() and len(a) > 100
    worst, close, total = 0.0, 0, 0
    for k in a:
        if not a[k].dtype.is_floating_point:
            assert torch.equal(a[k], b[k]), k
            continue
        d = (a[k] - b[k]).abs()
        worst = max(worst, float(d.max()))
        close += int((d <= 1e-5).sum())
        total += d.numel()
    assert worst <= 2 * iters * 5e-4 + 1e-6, worst            # lr 5e-4: a weight moves by at most lr per step
    assert close / total > 0.97, (close, total, worst)
