#!/usr/bin/env python3
"""The north-star multi-GPU job in one launch: every spatial cell trained on the rank that owns it, the container merged in the job,
the merged model evaluated image-parallel.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 mega-nerf_amd/tools/train_cells.py \\
        --config_file configs/mega-nerf/rubble.yaml --dataset_path $DATA --mask_path $MASKS --exp_name $EXP [reference flags ...]

replaces the reference's three hand-offs through the filesystem:
  * parscripts/run_8.txt:1-8 -- eight `train.py --cluster_mask_path $MASKS/$i --exp_name $EXP-$i` processes, one per GPU.  Here rank r
    trains the cells `assign_submodules(n_cells, world)[r]` (cell j on rank j % world: 8 cells on 8 GPUs = one each; 25 on 8 = 4,3,3,..),
    each through `Runner.train()` on its own cluster-masked pixels (`--cluster_mask_path`, runner.py:652-656) with private fg + bg
    weights and optimisers: NO collective while training.  Checkpoints land where the reference's would: `$EXP-<cell>/<version>/models/`.
  * scripts/merge_submodules.py:33-78 -- checkpoints collected from disk.  Here `merge.merge_in_job`: ONE all_gather (RCCL over xGMI) of the
    flat fp32 weight buffers straight from the trainers' memory; rank 0 writes the TorchScript container (`--output`, default $EXP-merged.pt).
  * runner.py:495-510 -- validation metrics through temp files.  Here `Runner` evaluates the container with image i on rank i % world and
    ONE all_reduce of the packed metric vector (`distributed.all_reduce_metrics`); every rank ends with the same totals.
`$MASKS` is the output of scripts/create_cluster_masks.py (params.pt + one directory per cell).  A rank that owns several cells (Building:
25 cells on 8 GPUs = 4,3,3,...) trains them SIDE BY SIDE through one plan -- `training.JointCells`: every cell keeps its own `Runner.train()`
loop, dataset, epochs, checkpoints and random streams, their iterations meet in one `mnr_train_step` call whose MLP launches carry all the
cells' rows (the cells fill each other's launch tails: 0.83 of the fp32-MFMA peak instead of 0.76 for one cell at a time, bench.py
`--submodules`).  `--sequential_cells` trains them one after the other instead (same batches, same random numbers).  Runs single-process too (world 1).  MNR_SHARE_GPU=1: all ranks on device 0 over gloo -- the code-path check on
a one-GPU box (RCCL refuses two ranks on one device); unmeasured on a multi-GPU node: the driver's 8-GPU runs use bench.py.
"""
import json
import os
import sys
from argparse import Namespace
from pathlib import Path

import torch
import torch.distributed as dist

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))

from mega_nerf.distributed import assign_submodules                  # noqa: E402
from mega_nerf.merge import merge_in_job, save_container              # noqa: E402
from mega_nerf.opts import get_opts_base                              # noqa: E402
from mega_nerf.runner import Runner                                   # noqa: E402


def _options() -> Namespace:
    parser = get_opts_base()
    parser.add_argument('--exp_name', type=str, required=True, help='experiment prefix: cell j trains in <exp_name>-<j>')
    parser.add_argument('--dataset_path', type=str, required=True)
    parser.add_argument('--mask_path', type=str, required=True, help='output directory of scripts/create_cluster_masks.py')
    parser.add_argument('--output', type=str, default=None, help='merged container (default <exp_name>-merged.pt)')
    parser.add_argument('--skip_eval', default=False, action='store_true')
    parser.add_argument('--sequential_cells', default=False, action='store_true',
                        help='a rank that owns several cells trains them one after the other instead of side by side in one plan')
    return parser.parse_args()


def main(hp: Namespace) -> None:
    rank, world = int(os.environ.get('RANK', 0)), int(os.environ.get('WORLD_SIZE', 1))
    share = bool(os.environ.get('MNR_SHARE_GPU'))
    if share:
        os.environ['LOCAL_RANK'] = '0'
    torch.cuda.set_device(int(os.environ.get('LOCAL_RANK', 0)))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group(backend='gloo' if share else 'nccl')          # nccl = RCCL on ROCm
    clustering = torch.load(Path(hp.mask_path) / 'params.pt', map_location='cpu', weights_only=False)
    n_cells = len(clustering['centroids'])
    mine = assign_submodules(n_cells, world)[rank]
    output = hp.output or '{}-merged.pt'.format(hp.exp_name)

    # ---- train: one independent single-process Runner per owned cell (the reference's one-process-per-cell layout); the cells of a rank
    # side by side in ONE plan (training.JointCells) unless --sequential_cells -------------------------------------------------------
    job_env = {k: os.environ.pop(k) for k in ('RANK', 'WORLD_SIZE') if k in os.environ}      # Runner: no DDP, this rank is its own master
    local = {}
    try:
        def make_runner(j):
            cell_hp = Namespace(**vars(hp))
            cell_hp.cluster_mask_path = str(Path(hp.mask_path) / str(j))
            cell_hp.exp_name = '{}-{}'.format(hp.exp_name, j)
            return Runner(cell_hp)
        if len(mine) > 1 and not hp.sequential_cells:
            from mega_nerf.training import JointCells
            runners = [make_runner(j) for j in mine]
            joint = JointCells(len(mine))
            for i, r in enumerate(runners):
                r.trainer_factory = joint.member(i)
            joint.run([r.train for r in runners])
            for j, r in zip(mine, runners):
                local[j] = (r.nerf, r.bg_nerf)
            print('rank {} trained cells {} side by side: {} joint steps, {} cell-by-cell iterations'.format(
                rank, mine, joint.joint_steps, joint.separate_steps), flush=True)
            del runners, joint
        else:
            for j in mine:
                runner = make_runner(j)
                runner.train()
                local[j] = (runner.nerf, runner.bg_nerf)
                print('rank {} trained cell {} ({} train images)'.format(rank, j, len(runner.train_items)), flush=True)
                del runner
    finally:
        os.environ.update(job_env)

    # ---- merge: one all_gather of the flat weight buffers, rank 0 writes the container ------------------------------------------
    device = torch.device('cpu') if share or world == 1 else torch.device('cuda', torch.cuda.current_device())
    container = merge_in_job(hp, local, clustering, device=device)
    if rank == 0:
        save_container(container, output)
    if world > 1:
        dist.barrier()
    if hp.skip_eval:
        return

    # ---- evaluate the merged container: image i on rank i % world, one all_reduce of the metric sums -----------------------------
    eval_hp = Namespace(**vars(hp))
    eval_hp.container_path, eval_hp.cluster_mask_path, eval_hp.ckpt_path = output, None, None
    eval_hp.exp_name = '{}-eval'.format(hp.exp_name)
    evaluator = Runner(eval_hp)
    evaluator._setup_experiment_dir()
    totals = evaluator._run_validation(0)
    evaluator._write_final_metrics(totals)
    n_val = max(1, len(evaluator.val_items))
    print('TRAIN_CELLS ' + json.dumps({'rank': rank, 'world': world, 'cells': mine, 'container': output,
                                       'val_psnr': totals['val/psnr'] / n_val, 'val_ssim': totals['val/ssim'] / n_val}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main(_options())
