#!/usr/bin/env python3
"""Minimal launch set for rocprofv3 --pmc passes: the fg training kernels at the benchmark's launch shapes
(131 072-row fine pass forward-with-tape + data-gradient chain, 196 608-row dense weight-gradient launch) and the fg
inference forward, a few repetitions each, nothing else in the process -- counter collection serialises every dispatch,
so profiling bench.py itself is slow.  Usage: rocprofv3 --pmc FETCH_SIZE WRITE_SIZE -d OUT -- python pmc_probe.py"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from mega_nerf.models.nerf import NeRF, ShiftedSoftplus   # noqa: E402


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    dev = torch.device('cuda')
    torch.manual_seed(0)
    m = NeRF(12, 4, 8, [4], 256, 48, False, 100, 3, 3, ShiftedSoftplus()).to(dev)
    for rows, S in ((131072, 128), (196608, 192)):
        n_rays = rows // S
        xyz = torch.rand(rows, 3, device=dev) * 2 - 1
        dirs = torch.nn.functional.normalize(torch.randn(n_rays, 3, device=dev), dim=-1)
        idx = torch.randint(0, 100, (n_rays,), device=dev).float()
        out = torch.empty(rows, 4, device=dev)
        d_out = torch.randn(rows, 4, device=dev)
        for _ in range(reps):
            with torch.no_grad():
                m.evaluate(xyz, 3, dirs, 3, idx, 1, S, rows, out)
            grads = {k: torch.zeros_like(p) for k, p in m.named_parameters()}
            tape = m.train_eval(xyz, 3, dirs, 3, S, idx, 1, S, rows, out, None, -1, None, 0)
            tape.backward(d_out, 4, grads)
        torch.cuda.synchronize()


if __name__ == '__main__':
    main()
