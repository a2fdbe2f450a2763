"""world_size-2 tests of the multi-GPU host logic on CPU (gloo): metric all-reduce, submodule weight gather,
submodule / image assignment.  The same code runs over RCCL on the GPU node (bench.py --gpus N)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from mega_nerf import distributed as D


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        # metric reduce: rank r evaluated images r, r+2, ... with psnr = 20 + image index
        imgs = D.images_for_rank(5, rank, world)
        sums, n = D.all_reduce_metrics({'val/psnr': float(sum(20 + i for i in imgs)), 'val/ssim': 0.5 * len(imgs)},
                                       len(imgs), torch.device('cpu'))
        assert n == 5 and abs(sums['val/psnr'] - sum(20 + i for i in range(5))) < 1e-9 and abs(sums['val/ssim'] - 2.5) < 1e-9
        # weight gather: every rank owns a different "submodule"
        g = torch.Generator().manual_seed(100 + rank)
        state = {'sigma.weight': torch.randn(1, 8, generator=g), 'rgb.bias': torch.randn(3, generator=g),
                 'xyz_encodings.0.0.weight': torch.randn(8, 5, generator=g)}
        allw = D.gather_submodule_weights(state)
        assert len(allw) == world
        for r in range(world):
            gr = torch.Generator().manual_seed(100 + r)
            exp = {'sigma.weight': torch.randn(1, 8, generator=gr), 'rgb.bias': torch.randn(3, generator=gr),
                   'xyz_encodings.0.0.weight': torch.randn(8, 5, generator=gr)}
            for k in exp:
                assert torch.equal(allw[r][k], exp[k]), (r, k)
        # data-parallel gradient averaging (Runner.train with several ranks on one submodule): two replicas with different
        # "local" gradients, one parameter without a gradient on rank 1 -> identical means everywhere
        torch.manual_seed(7)
        lin = torch.nn.Linear(4, 3)
        lin.weight.grad = torch.full_like(lin.weight, float(rank + 1))
        lin.bias.grad = torch.arange(3.) * (rank + 1) if rank == 0 else None
        D.average_gradients(list(lin.parameters()))
        assert torch.allclose(lin.weight.grad, torch.full_like(lin.weight, 1.5))
        assert torch.allclose(lin.bias.grad, torch.arange(3.) * 0.5)
        # gradients that are views of one flat buffer (the fused step's gradient area): reduced in place, padding included
        flat = torch.zeros(4 * 3 + 4 + 3)
        lin.weight.grad = flat[0:12].view(3, 4)
        lin.bias.grad = flat[16:19]
        flat[0:12] = float(rank + 1)
        flat[16:19] = torch.arange(3.) * (rank + 1)
        assert D._as_one_buffer([lin.weight.grad, lin.bias.grad]) is not None
        D.average_gradients(list(lin.parameters()))
        assert lin.weight.grad.data_ptr() == flat.data_ptr() and torch.allclose(flat[0:12], torch.full((12,), 1.5))
        assert torch.allclose(flat[16:19], torch.arange(3.) * 1.5)
        assert D.any_rank(rank == 1, torch.device('cpu')) is True and D.any_rank(False, torch.device('cpu')) is False
        # strong-scaling cell -> rank -> batch mapping of bench.py --submodules 8: every cell exactly once over the ranks,
        # seeds independent of the world size
        mine = D.assign_submodules(8, world)[rank]
        cells = [None] * world
        dist.all_gather_object(cells, mine)
        assert sorted(c for part in cells for c in part) == list(range(8)) and all(c % world == rank for c in mine)
        dist.barrier()
        out.put((rank, 'ok'))
    except Exception as e:      # pragma: no cover
        out.put((rank, repr(e)))
    finally:
        dist.destroy_process_group()


def test_two_rank_metric_reduce_and_weight_gather():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, 'ok'), (1, 'ok')], res


def test_assignment():
    a = D.assign_submodules(25, 8)
    assert [len(x) for x in a] == [4, 3, 3, 3, 3, 3, 3, 3]
    assert sorted(j for x in a for j in x) == list(range(25))
    assert D.assign_submodules(8, 8) == [[j] for j in range(8)]
    assert D.images_for_rank(5, 1, 2) == [1, 3]


def test_single_process_paths():
    sums, n = D.all_reduce_metrics({'a': 1.5}, 3, torch.device('cpu'))
    assert sums == {'a': 1.5} and n == 3
    st = {'w': torch.arange(6.).reshape(2, 3)}
    out = D.gather_submodule_weights(st)
    assert len(out) == 1 and torch.equal(out[0]['w'], st['w'])
