"""Runner -- training / evaluation driver with the reference's entry points and on-disk formats
(reference: mega_nerf/runner.py:38-673), driving the MI355X kernels.

What is kept: constructor semantics (coordinates.pt, near/far/altitude normalisation, cluster-mask parameter
checks, ellipsoid bounds from the camera extents), ``train()`` / ``eval()`` / ``render_image()`` /
``_training_step()`` signatures, Adam + ExponentialLR, the checkpoint dictionary keys (runner.py:521-536), the
experiment directory layout (``<exp>/<version>/{hparams.txt,command.txt,image_indices.txt,models/,metrics.txt}``),
right-half validation PSNR, image-parallel validation (image i on rank i % world).
What changes (MI355X-first): the training set is device resident and batches are drawn on the device; compute is
fp32 (``--no_amp`` is implied); validation metrics are combined with ONE all_reduce instead of temp files
(mega_nerf.distributed); SSIM/LPIPS/TensorBoard/JPEG dumps are out of scope (optional if the packages exist).
Training of the default architecture runs the whole iteration (render, loss, backward, Adam on both models, re-pack) as ONE
native call per step (``mnr_train_step`` through ``training.CellTrainer``) with no host synchronisation between the log /
checkpoint intervals; the optimiser objects and the checkpoint's ``optimizers`` entry stay torch.optim.Adam's.
The multi-GPU layout is one submodule per GPU (parscripts/run_8.txt), i.e. independent single-rank Runners; the reference's
DDP mode (several ranks on one submodule, runner.py:120-129) is kept on the stage-by-stage autograd path.
"""
from __future__ import annotations

import math
import os
import random
import sys
from argparse import Namespace
from collections import defaultdict
from pathlib import Path
from typing import Dict, List, Optional, Tuple, Union

import numpy as np
import torch
import torch.distributed as dist
import torch.nn.functional as F
from torch.optim import Adam
from torch.optim.lr_scheduler import ExponentialLR

from mega_nerf import distributed as mdist
from mega_nerf.datasets.memory_dataset import MemoryDataset
from mega_nerf.image_metadata import ImageMetadata
from mega_nerf.metrics import psnr, psnr_ssim
from mega_nerf.misc_utils import main_print, main_tqdm
from mega_nerf.models.model_utils import get_bg_nerf, get_nerf
from mega_nerf.ray_utils import get_ray_directions, get_rays
from mega_nerf.rendering import render_rays


class Runner:
    def __init__(self, hparams: Namespace, set_experiment_path: bool = True):
        if hparams.ckpt_path is not None:
            ckpt = torch.load(hparams.ckpt_path, map_location='cpu', weights_only=False)
            np.random.set_state(ckpt['np_random_state'])
            torch.set_rng_state(ckpt['torch_random_state'])
            random.setstate(ckpt['random_state'])
        else:
            np.random.seed(hparams.random_seed)
            torch.manual_seed(hparams.random_seed)
            random.seed(hparams.random_seed)
        self.hparams = hparams

        self.distributed = 'RANK' in os.environ and int(os.environ.get('WORLD_SIZE', 1)) > 1
        if self.distributed and not dist.is_initialized():
            os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
            dist.init_process_group(backend='nccl')          # RCCL on ROCm
        if torch.cuda.is_available():
            torch.cuda.set_device(int(os.environ.get('LOCAL_RANK', 0)))
        self.is_master = int(os.environ.get('RANK', 0)) == 0
        self.is_local_master = int(os.environ.get('LOCAL_RANK', 0)) == 0
        main_print(hparams)

        if set_experiment_path:
            self.experiment_path = self._get_experiment_path() if self.is_master else None
            self.model_path = self.experiment_path / 'models' if self.is_master else None
        self.writer = None
        self.iteration_hook = None        # optional callable(train_iterations), called after every training iteration (bench.py times the loop with it)
        if not torch.cuda.is_available():
            raise RuntimeError('mega_nerf (MI355X build) needs a HIP device: there is no CPU fallback')
        self.device = torch.device('cuda', torch.cuda.current_device())

        coords = torch.load(Path(hparams.dataset_path) / 'coordinates.pt', map_location='cpu', weights_only=False)
        self.origin_drb = coords['origin_drb']
        self.pose_scale_factor = coords['pose_scale_factor']
        main_print('Origin: {}, scale factor: {}'.format(self.origin_drb, self.pose_scale_factor))
        self.near = hparams.near / self.pose_scale_factor
        if hparams.far is not None:
            self.far = hparams.far / self.pose_scale_factor
        else:
            self.far = 1e5 if hparams.bg_nerf else 2
        main_print('Ray bounds: {}, {}'.format(self.near, self.far))
        self.ray_altitude_range = [float((x - self.origin_drb[0]) / self.pose_scale_factor)
                                   for x in hparams.ray_altitude_range] if hparams.ray_altitude_range is not None else None
        main_print('Ray altitude range in [-1, 1] space: {}'.format(self.ray_altitude_range))
        if self.ray_altitude_range is not None:
            assert self.ray_altitude_range[0] < self.ray_altitude_range[1]

        if hparams.cluster_mask_path is not None:
            cp = torch.load(Path(hparams.cluster_mask_path).parent / 'params.pt', map_location='cpu', weights_only=False)
            assert cp['near'] == self.near
            assert torch.allclose(cp['origin_drb'], self.origin_drb)
            assert cp['pose_scale_factor'] == self.pose_scale_factor
            if self.ray_altitude_range is not None:
                assert torch.allclose(torch.FloatTensor(cp['ray_altitude_range']), torch.FloatTensor(self.ray_altitude_range))

        self.train_items, self.val_items = self._get_image_metadata()
        main_print('Using {} train images and {} val images'.format(len(self.train_items), len(self.val_items)))
        cams = torch.stack([x.c2w[:3, 3] for x in self.train_items + self.val_items])
        min_position, max_position = cams.min(dim=0)[0], cams.max(dim=0)[0]

        self.nerf = get_nerf(hparams, len(self.train_items)).to(self.device)
        self.bg_nerf = None
        self.sphere_center = self.sphere_radius = None
        if hparams.bg_nerf:
            self.bg_nerf = get_bg_nerf(hparams, len(self.train_items)).to(self.device)
            if hparams.ellipse_bounds:
                assert hparams.ray_altitude_range is not None
                ground, air = cams.clone(), cams.clone()
                ground[:, 0], air[:, 0] = self.ray_altitude_range[1], self.ray_altitude_range[0]
                used = torch.cat([cams, air, ground])
                max_position[0] = self.ray_altitude_range[1]
                self.sphere_center = ((max_position + min_position) * 0.5).to(self.device)
                self.sphere_radius = ((max_position - min_position) * 0.5).to(self.device)
                scale = ((used.to(self.device) - self.sphere_center) / self.sphere_radius).norm(dim=-1).max()
                self.sphere_radius = self.sphere_radius * (scale * hparams.ellipse_scale_factor)
            main_print('Sphere center: {}, radius: {}'.format(self.sphere_center, self.sphere_radius))

    # ------------------------------------------------------------------------------------------------
    def train(self):
        # Several ranks on ONE submodule (the reference's DDP + DistributedSampler mode, runner.py:120-129,228-238): every rank
        # walks the same shuffled epoch and trains on the batches  index % world == rank; gradients are averaged with ONE
        # all_reduce over a flat buffer before the optimiser steps, so all ranks hold identical weights.  (The Mega-NeRF layout
        # proper -- one submodule per GPU, parscripts/run_8.txt -- is N independent single-rank runs and needs none of this.)
        world = dist.get_world_size() if self.distributed else 1
        rank = dist.get_rank() if self.distributed else 0
        hp = self.hparams
        self._setup_experiment_dir()
        optimizers = {'nerf': Adam(self.nerf.parameters(), lr=hp.lr)}
        if self.bg_nerf is not None:
            optimizers['bg_nerf'] = Adam(self.bg_nerf.parameters(), lr=hp.lr)
        train_iterations = 0
        epoch, discard = 0, 0
        if hp.ckpt_path is not None:
            ckpt = torch.load(hp.ckpt_path, map_location='cpu', weights_only=False)
            train_iterations = ckpt['iteration']
            if hp.resume_ckpt_state:
                # resume inside the epoch the checkpoint was taken in: same permutation (seeded by the epoch number), the
                # batches already consumed are skipped (the reference's discard_index, runner.py:213-226)
                epoch, discard = int(ckpt.get('epoch', 0)), int(ckpt.get('dataset_index', -1)) + 1
                discard = -(-discard // world) * world         # rank 0's index closes a group of `world` batches: resume at the next group
            for key, opt in optimizers.items():
                sd = opt.state_dict()
                sd.update(ckpt['optimizers'][key])
                opt.load_state_dict(sd)
        schedulers = {k: ExponentialLR(o, gamma=hp.lr_decay_factor ** (1 / hp.train_iterations),
                                       last_epoch=train_iterations - 1) for k, o in optimizers.items()}
        # One rank per submodule (the Mega-NeRF layout): the iteration is ONE native call (training.CellTrainer -> mnr_train_step)
        # whenever the configuration has a fused step, on the SAME optimiser / scheduler objects (their moment tensors become views
        # of the step's buffers, so checkpoints keep the reference's `optimizers` entry); anything else (512-wide cells, other sample
        # counts, ragged last batches) takes the stage-by-stage autograd path inside the same trainer, whose per-iteration host checks
        # ride behind the forward pass instead of idling the GPU twice per step.  MNR_RUNNER_AUTOGRAD=1 keeps the reference-shaped loop below.
        from mega_nerf.training import CellTrainer, GatheredBatch
        trainer = None
        if world == 1 and hp.appearance_dim > 0 and not os.environ.get('MNR_RUNNER_AUTOGRAD'):
            # (trainer_factory: tools/train_cells.py hands the cells of one rank members of a training.JointCells group -- one plan for all)
            make = getattr(self, 'trainer_factory', None) or CellTrainer
            trainer = make(self.nerf, self.bg_nerf, hp, self.sphere_center, self.sphere_radius, optimizers, schedulers,
                           seed=int(hp.random_seed), iteration=train_iterations, plan_rays=int(hp.batch_size))
        self.trainer = trainer
        check_every = max(1, min(hp.ckpt_interval, 100))      # fused path: loss finiteness / sphere errors are checked at this interval
        filesystem = hp.dataset_type == 'filesystem'
        chunk_ready = False
        if filesystem:
            # chunked on-disk dataset (runner.py:196-209): same chunk files / checkpoint state as the reference
            from mega_nerf.datasets.filesystem_dataset import FilesystemDataset
            if hp.chunk_paths is None:
                raise Exception('--chunk_paths is required for --dataset_type filesystem')
            dataset = FilesystemDataset(self.train_items, self.near, self.far, self.ray_altitude_range, hp.center_pixels,
                                        self.device, [Path(x) for x in sorted(hp.chunk_paths)], hp.num_chunks,
                                        hp.train_scale_factor, hp.disk_flush_size)
            if hp.ckpt_path is not None and hp.resume_ckpt_state and 'dataset_state' in ckpt:
                dataset.set_state(ckpt['dataset_state'])      # resume inside the chunk the checkpoint was taken in
                chunk_ready = True
        else:
            dataset = MemoryDataset(self.train_items, self.near, self.far, self.ray_altitude_range, hp.center_pixels,
                                    self.device)
        self.nerf.train()
        if self.bg_nerf is not None:
            self.bg_nerf.train()
        dataset_index = 0
        while train_iterations < hp.train_iterations:
            if filesystem and not chunk_ready:
                dataset.load_chunk()                      # next chunk (prefetched on its own stream by a worker thread)
            chunk_ready = False
            gen = torch.Generator().manual_seed(int(hp.random_seed) + 1000003 * epoch)     # same shuffle on every rank / after a resume
            # data-parallel ranks walk the epoch in groups of `world` consecutive batches (one each) and drop the ragged last group:
            # every rank then takes the same number of steps per epoch and their collectives pair up
            usable = (-(-len(dataset) // hp.batch_size) // world) * world
            if usable == 0:
                raise Exception('{} training pixels give fewer batches of {} than there are ranks ({}): nothing to train on'.format(
                    len(dataset), hp.batch_size, world))
            # the one-call step gathers its batch itself from the resident arrays (training.GatheredBatch): the loop then enqueues NO torch
            # kernel per iteration; every other path gets materialised batches
            source = dataset.gather_source() if trainer is not None else None
            if trainer is not None and trainer.fused is None:
                # a cell whose usable pixels are fewer than --batch_size (small / cluster-masked cells, test datasets) takes every batch at
                # that smaller size: plan the one-call step for it instead of never fusing
                trainer.plan_rays = min(int(hp.batch_size), len(dataset))
            # the epoch as row selections: a rank materialises (gathers) only the batches it trains on
            for dataset_index, item in enumerate(dataset.index_batches(hp.batch_size, gen)):
                if dataset_index < discard or dataset_index >= usable or dataset_index % world != rank:
                    continue
                if trainer is not None:
                    loss_dev, _, _ = trainer.step_gathered(GatheredBatch(source[0], source[1], source[2], item, source[3]))
                    train_iterations += 1
                    last = train_iterations >= hp.train_iterations
                    if train_iterations % check_every == 0 or last or train_iterations % hp.ckpt_interval == 0:
                        trainer.health()                      # raises what the reference raises per iteration (runner.py:260-261)
                        loss = float(loss_dev)
                        if not math.isfinite(loss):
                            raise Exception('Train metrics not finite: {}'.format({'loss': loss}))
                        if self.is_master:
                            # PSNR of rgb_fine as the reference logs it (runner.py:252-256); under --use_cascade the loss is the mean of the
                            # fine and the coarse MSE, so it comes from the trainer's own fine-pass MSE, not from the loss
                            mse = float(getattr(trainer, 'last_mse', loss_dev))
                            main_print('iter {}: psnr {:.3f} loss {:.5f}'.format(train_iterations, -10 * math.log10(max(mse, 1e-30)), loss))
                    if self.is_master and train_iterations % hp.ckpt_interval == 0:
                        trainer.sync()
                        self._save_checkpoint(optimizers, None, train_iterations, dataset_index,
                                              dataset.get_state() if filesystem else None, epoch)
                    if train_iterations % hp.val_interval == 0:
                        self._run_validation(train_iterations)
                    if self.iteration_hook is not None:
                        self.iteration_hook(train_iterations)
                    if last:
                        break
                    continue
                item = dataset[item]
                image_indices = item['img_indices'] if hp.appearance_dim > 0 else None
                metrics, bg_present = self._training_step(item['rgbs'], item['rays'], image_indices)
                for key, val in metrics.items():
                    val = float(val.detach()) if isinstance(val, torch.Tensor) else float(val)
                    if key == 'psnr' and math.isinf(val):
                        continue
                    if not math.isfinite(val):
                        raise Exception('Train metrics not finite: {}'.format(metrics))
                for opt in optimizers.values():
                    opt.zero_grad(set_to_none=True)
                metrics['loss'].backward()
                if world > 1:
                    from mega_nerf.distributed import any_rank, average_gradients
                    bg_present = any_rank(bg_present, self.device)
                    average_gradients([p for opt in optimizers.values() for group in opt.param_groups for p in group['params']])
                for key, opt in optimizers.items():
                    if key == 'bg_nerf' and not bg_present:
                        continue
                    opt.step()
                for sch in schedulers.values():
                    sch.step()
                train_iterations += 1
                if self.is_master and train_iterations % max(1, min(hp.ckpt_interval, 100)) == 0:
                    main_print('iter {}: psnr {:.3f} loss {:.5f}'.format(train_iterations, metrics['psnr'],
                                                                        float(metrics['loss'].detach())))
                if self.is_master and train_iterations % hp.ckpt_interval == 0:
                    self._save_checkpoint(optimizers, None, train_iterations, dataset_index,
                                          dataset.get_state() if filesystem else None, epoch)
                if train_iterations % hp.val_interval == 0:
                    self._run_validation(train_iterations)
                if self.iteration_hook is not None:
                    self.iteration_hook(train_iterations)
                if train_iterations >= hp.train_iterations:
                    break
            else:
                epoch, discard = epoch + 1, 0           # the epoch ran to its end
                continue
            break
        if trainer is not None:
            trainer.sync()
        if self.is_master:
            self._save_checkpoint(optimizers, None, train_iterations, dataset_index, dataset.get_state() if filesystem else None, epoch)
        if hp.cluster_mask_path is None:
            self._write_final_metrics(self._run_validation(train_iterations))

    def eval(self):
        self._setup_experiment_dir()
        self._write_final_metrics(self._run_validation(0))

    def _write_final_metrics(self, val_metrics: Dict[str, float]) -> None:
        if self.is_master:
            with (self.experiment_path / 'metrics.txt').open('w') as f:
                for key in val_metrics:
                    message = 'Average {}: {}'.format(key, val_metrics[key] / max(1, len(self.val_items)))
                    main_print(message)
                    f.write('{}\n'.format(message))

    def _setup_experiment_dir(self) -> None:
        if self.is_master:
            self.experiment_path.mkdir()
            with (self.experiment_path / 'hparams.txt').open('w') as f:
                for key, val in vars(self.hparams).items():
                    f.write('{}: {}\n'.format(key, val))
                if 'WORLD_SIZE' in os.environ:
                    f.write('WORLD_SIZE: {}\n'.format(os.environ['WORLD_SIZE']))
            (self.experiment_path / 'command.txt').write_text(' '.join(sys.argv) + '\n')
            self.model_path.mkdir(parents=True)
            with (self.experiment_path / 'image_indices.txt').open('w') as f:
                for item in self.train_items:
                    f.write('{},{}\n'.format(item.image_index, item.image_path.name))
        if self.distributed:
            dist.barrier()

    def _training_step(self, rgbs: torch.Tensor, rays: torch.Tensor, image_indices: Optional[torch.Tensor]) \
            -> Tuple[Dict[str, Union[torch.Tensor, float]], bool]:
        results, bg_present = render_rays(nerf=self.nerf, bg_nerf=self.bg_nerf, rays=rays, image_indices=image_indices,
                                          hparams=self.hparams, sphere_center=self.sphere_center,
                                          sphere_radius=self.sphere_radius, get_depth=False, get_depth_variance=True,
                                          get_bg_fg_rgb=False)
        typ = 'fine' if 'rgb_fine' in results else 'coarse'
        with torch.no_grad():
            metrics = {'psnr': psnr(results[f'rgb_{typ}'], rgbs),
                       'depth_variance': results[f'depth_variance_{typ}'].mean()}
        photo_loss = F.mse_loss(results[f'rgb_{typ}'], rgbs, reduction='mean')
        metrics['photo_loss'] = photo_loss
        metrics['loss'] = photo_loss
        if self.hparams.use_cascade and typ != 'coarse':
            coarse_loss = F.mse_loss(results['rgb_coarse'], rgbs, reduction='mean')
            metrics['coarse_loss'] = coarse_loss
            metrics['loss'] = (photo_loss + coarse_loss) / 2
        return metrics, bg_present

    def _run_validation(self, train_index: int) -> Dict[str, float]:
        """PSNR and SSIM over the right half of every validation image (runner.py:413-436), evaluated on the device in one
        pass per image (LPIPS is out of scope); images are split over the ranks and the sums combined with one all_reduce."""
        world = int(os.environ.get('WORLD_SIZE', 1)) if self.distributed else 1
        rank = int(os.environ.get('RANK', 0)) if self.distributed else 0
        sums = defaultdict(float)
        with torch.inference_mode():
            was_training = self.nerf.training
            self.nerf.eval()          # NB the reference leaves bg_nerf in training mode here (random, unsorted
            bg_was = self.bg_nerf.training if self.bg_nerf is not None else False     # bg fine samples + sigma noise at validation);
            if self.bg_nerf is not None and not os.environ.get('MNR_REFERENCE_BG_EVAL_MODE'):
                self.bg_nerf.eval()   # default: both deterministic (scripts/render_images.py:73-75); MNR_REFERENCE_BG_EVAL_MODE=1 = the quirk
            count = 0
            for i in main_tqdm(mdist.images_for_rank(len(self.val_items), rank, world)):
                item = self.val_items[i]
                gt = (item.load_image().float() / 255.).to(self.device)
                results, _ = self.render_image(item)
                typ = 'fine' if 'rgb_fine' in results else 'coarse'
                pred = results[f'rgb_{typ}'].view(*gt.shape)
                half = gt.shape[1] // 2
                val_psnr, val_ssim = psnr_ssim(pred[:, half:], gt[:, half:], 1.0)     # strided right-half views, no copy
                sums['val/psnr'] += val_psnr
                sums['val/ssim'] += val_ssim
                count += 1
                self._save_validation_images(train_index, i, gt, results, typ)
            self.nerf.train(was_training)
            if self.bg_nerf is not None:
                self.bg_nerf.train(bg_was)
        # validation renders 65 536-ray chunks: the one-call render's scratch (~2.7 GB) and, for containers / --train_mega_nerf, the routing
        # buffers of both models (n_sub x rows x 28 bytes + the cells' outputs per kept size: GBs) must not stay pinned while training resumes
        from mega_nerf.rendering import release_render_workspaces
        release_render_workspaces(self.nerf, self.bg_nerf)
        sums.setdefault('val/psnr', 0.0)
        sums.setdefault('val/ssim', 0.0)
        total, _ = mdist.all_reduce_metrics(dict(sums), count, self.device)
        return total

    def _save_validation_images(self, train_index: int, i: int, gt: torch.Tensor, results: Dict[str, torch.Tensor], typ: str) -> None:
        """ground truth | render | log-depth panels of validation image ``i`` (+ the background / foreground panels), the images the
        reference hands to its TensorBoard writer (runner.py:452-491); there is no TensorBoard here, so the rank that owns the experiment
        directory writes them to <experiment>/val_images/<iteration>/.  MNR_NO_VAL_IMAGES=1 skips it."""
        path = getattr(self, 'experiment_path', None)
        if path is None or os.environ.get('MNR_NO_VAL_IMAGES'):
            return
        out = path / 'val_images' / str(train_index)
        out.mkdir(parents=True, exist_ok=True)
        H, W = gt.shape[0], gt.shape[1]

        def panel(rgb_key: str, depth_key: str, clamp_to: Optional[str], name: str) -> None:
            depth = torch.nan_to_num(results[depth_key]).view(-1)
            if clamp_to is not None and clamp_to in results:      # background depths are inverse-sphere quantities (quirk Q2): clamp for display
                to_use = torch.nan_to_num(results[clamp_to]).view(-1)
                while to_use.shape[0] > 2 ** 24:
                    to_use = to_use[::2]
                depth = depth.clamp_max(torch.quantile(to_use, 0.95))
            Runner._create_result_image(gt, results[rgb_key].view(H, W, 3), depth).save(str(out / name))

        panel(f'rgb_{typ}', f'depth_{typ}', f'fg_depth_{typ}', '{}.jpg'.format(i))
        if self.hparams.bg_nerf and f'bg_rgb_{typ}' in results:
            panel(f'bg_rgb_{typ}', f'bg_depth_{typ}', None, '{}_bg.jpg'.format(i))
            panel(f'fg_rgb_{typ}', f'fg_depth_{typ}', None, '{}_fg.jpg'.format(i))

    def _save_checkpoint(self, optimizers: Dict[str, any], scaler, train_index: int, dataset_index: int,
                         dataset_state: Optional[str], epoch: int = 0) -> None:
        ckpt = {
            'model_state_dict': self.nerf.state_dict(),
            'scaler': {},                                       # fp32 compute: no GradScaler state
            'optimizers': {k: v.state_dict() for k, v in optimizers.items()},
            'iteration': train_index,
            'torch_random_state': torch.get_rng_state(),
            'np_random_state': np.random.get_state(),
            'random_state': random.getstate(),
            'dataset_index': dataset_index,
            'epoch': epoch,                                      # (extra key) which shuffle `dataset_index` counts in
        }
        if dataset_state is not None:
            ckpt['dataset_state'] = dataset_state
        if self.bg_nerf is not None:
            ckpt['bg_model_state_dict'] = self.bg_nerf.state_dict()
        torch.save(ckpt, self.model_path / '{}.pt'.format(train_index))

    def render_image(self, metadata: ImageMetadata) -> Tuple[Dict[str, torch.Tensor], torch.Tensor]:
        hp = self.hparams
        directions = get_ray_directions(metadata.W, metadata.H, metadata.intrinsics[0], metadata.intrinsics[1],
                                        metadata.intrinsics[2], metadata.intrinsics[3], hp.center_pixels, self.device)
        rays = get_rays(directions, metadata.c2w.to(self.device), self.near, self.far, self.ray_altitude_range).view(-1, 8)
        image_indices = metadata.image_index * torch.ones(rays.shape[0], device=rays.device) \
            if hp.appearance_dim > 0 else None
        chunks = defaultdict(list)
        for i in range(0, rays.shape[0], hp.image_pixel_batch_size):
            batch, _ = render_rays(nerf=self.nerf, bg_nerf=self.bg_nerf, rays=rays[i:i + hp.image_pixel_batch_size],
                                   image_indices=image_indices[i:i + hp.image_pixel_batch_size]
                                   if image_indices is not None else None,
                                   hparams=hp, sphere_center=self.sphere_center, sphere_radius=self.sphere_radius,
                                   get_depth=True, get_depth_variance=False, get_bg_fg_rgb=True)
            for key, value in batch.items():
                chunks[key].append(value)
        return {k: torch.cat(v) for k, v in chunks.items()}, rays

    def _get_image_metadata(self) -> Tuple[List[ImageMetadata], List[ImageMetadata]]:
        root = Path(self.hparams.dataset_path)
        cand = sorted((root / 'train' / 'metadata').iterdir())
        train_paths = [cand[i] for i in range(0, len(cand), self.hparams.train_every)]
        val_paths = sorted((root / 'val' / 'metadata').iterdir())
        train_paths = sorted(train_paths + val_paths, key=lambda x: x.name)
        val_set = set(val_paths)
        index_of = {p.name: i for i, p in enumerate(train_paths)}
        train_items = [self._get_metadata_item(p, index_of[p.name], self.hparams.train_scale_factor, p in val_set)
                       for p in train_paths]
        val_items = [self._get_metadata_item(p, index_of[p.name], self.hparams.val_scale_factor, True) for p in val_paths]
        return train_items, val_items

    def _get_metadata_item(self, metadata_path: Path, image_index: int, scale_factor: int, is_val: bool) -> ImageMetadata:
        image_path = None
        for ext in ('.jpg', '.JPG', '.png', '.PNG'):
            c = metadata_path.parent.parent / 'rgbs' / (metadata_path.stem + ext)
            if c.exists():
                image_path = c
                break
        assert image_path is not None and image_path.exists()
        md = torch.load(metadata_path, map_location='cpu', weights_only=False)
        assert md['W'] % scale_factor == 0 and md['H'] % scale_factor == 0
        dataset_mask = metadata_path.parent.parent.parent / 'masks' / metadata_path.name
        if self.hparams.cluster_mask_path is not None:
            mask_path = Path(self.hparams.cluster_mask_path) / metadata_path.name
        elif dataset_mask.exists():
            mask_path = dataset_mask
        else:
            mask_path = None
        return ImageMetadata(image_path, md['c2w'], md['W'] // scale_factor, md['H'] // scale_factor,
                             md['intrinsics'] / scale_factor, image_index,
                             None if (is_val and self.hparams.all_val) else mask_path, is_val)

    # anchor colours of the depth ramp (dark violet -> magenta -> orange -> pale yellow, 0 = dark); the reference maps through OpenCV's
    # COLORMAP_INFERNO table (runner.py:610) -- OpenCV is not part of this image, so the ramp is piecewise linear through these anchors:
    # same ordering and endpoints, colours differ by a few counts in between (visualisation only, nothing reads these images back)
    _RAMP = np.array([[0, 0, 4], [40, 11, 84], [101, 21, 110], [159, 42, 99], [212, 72, 66], [245, 125, 21], [250, 193, 39], [252, 255, 164]],
                     dtype=np.float32)

    @staticmethod
    def visualize_scalars(scalar_tensor: torch.Tensor) -> np.ndarray:
        """(H, W) scalars -> (H, W, 3) uint8 heat map: normalised between the 5 % and 95 % quantiles, inverted (near = bright), as
        runner.py:598-610 does for the depth panels."""
        to_use = scalar_tensor.reshape(-1).float()
        while to_use.shape[0] > 2 ** 24:
            to_use = to_use[::2]
        mi, ma = torch.quantile(to_use, 0.05), torch.quantile(to_use, 0.95)
        t = ((scalar_tensor.float() - mi) / max(float(ma - mi), 1e-8)).clamp(0, 1)
        level = ((1 - t) * 255).byte().cpu().numpy().astype(np.float32) / 255.0
        ramp = Runner._RAMP
        x = level * (len(ramp) - 1)
        i0 = np.minimum(x.astype(np.int64), len(ramp) - 2)
        f = (x - i0)[..., None]
        return (ramp[i0] * (1 - f) + ramp[i0 + 1] * f + 0.5).astype(np.uint8)

    @staticmethod
    def _create_result_image(rgbs: torch.Tensor, result_rgbs: torch.Tensor, result_depths: torch.Tensor):
        """ground truth | render | log-depth heat map side by side (runner.py:591-595)."""
        from PIL import Image
        depth_vis = Runner.visualize_scalars(torch.log(result_depths + 1e-8).view(rgbs.shape[0], rgbs.shape[1]).cpu())
        images = ((rgbs * 255).cpu().numpy(), (result_rgbs * 255).cpu().numpy(), depth_vis)
        return Image.fromarray(np.concatenate(images, 1).astype(np.uint8))

    def _get_experiment_path(self) -> Path:
        exp_dir = Path(self.hparams.exp_name)
        exp_dir.mkdir(parents=True, exist_ok=True)
        versions = [int(x.name) for x in exp_dir.iterdir() if x.name.isdigit()]
        return exp_dir / str(0 if not versions else max(versions) + 1)


# ---- command-line entry shared by train.py and eval.py ------------------------------------------------------------
def cli_options(argv: Optional[List[str]] = None) -> Namespace:
    """The reference's train/eval flag set: opts.get_opts_base() plus --exp_name and --dataset_path."""
    from mega_nerf.opts import get_opts_base
    parser = get_opts_base()
    for flag, text in (('--exp_name', 'experiment name'), ('--dataset_path', 'dataset root (train/, val/, coordinates.pt)')):
        parser.add_argument(flag, type=str, required=True, help=text)
    return parser.parse_args(argv)


def run_cli(hparams: Namespace, action: str) -> None:
    """Run ``Runner(hparams).train()`` / ``.eval()``, under autograd anomaly detection when --detect_anomalies is set."""
    from contextlib import nullcontext
    guard = torch.autograd.detect_anomaly() if hparams.detect_anomalies else nullcontext()
    with guard:
        getattr(Runner(hparams), action)()
