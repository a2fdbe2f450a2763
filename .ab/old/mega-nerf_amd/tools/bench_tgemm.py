"""Throughput of the tiled GEMM (csrc/tgemm.hip) on the layer shapes of the wide architectures, beside the 128 x 128
kernels it replaces (mnr_linear / mnr_gemm of csrc/layerwise.hip).  Prints one JSON line per shape."""
import ctypes as C
import json
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from mega_nerf import _native as N  # noqa: E402

DEV = 'cuda:0'


def timed(fn, reps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    lib = N.lib()
    st = N.stream_ptr
    shapes = [(196608, 512, 512), (196608, 256, 256), (65536, 2048, 2048), (196608, 256, 512), (100000, 512, 512)]
    if len(sys.argv) > 1:
        shapes = [tuple(int(v) for v in a.split('x')) for a in sys.argv[1:]]
    for M, Nn, K in shapes:
        x = torch.randn(M, K, device=DEV)
        w = torch.randn(Nn, K, device=DEV) / K ** 0.5
        b = torch.randn(Nn, device=DEV)
        y = torch.empty(M, Nn, device=DEV)
        h = torch.randn(M, K, device=DEV)
        dx = torch.empty(M, K, device=DEV)
        g = N.TGemm()
        g.a[0], g.lda[0], g.b[0], g.ldb[0], g.k[0] = x.data_ptr(), K, w.data_ptr(), K, K
        g.n_phases, g.b_kslow, g.relu, g.c, g.ldc, g.m, g.n, g.bias = 1, 0, 1, y.data_ptr(), Nn, M, Nn, b.data_ptr()
        d = N.TGemm()         # dX[M][K] = Y[M][Nn] . W[Nn][K], gated by h
        d.a[0], d.lda[0], d.b[0], d.ldb[0], d.k[0] = y.data_ptr(), Nn, w.data_ptr(), K, Nn
        d.n_phases, d.b_kslow, d.c, d.ldc, d.m, d.n, d.gate, d.ldgate = 1, 1, dx.data_ptr(), K, M, K, h.data_ptr(), K
        fl = 2.0 * M * Nn * K
        reps = max(3, int(2e12 / fl))
        t_new = timed(lambda: N.check(lib.mnr_tgemm_run(C.byref(g), st())), reps)
        t_old = timed(lambda: N.check(lib.mnr_linear(y.data_ptr(), Nn, x.data_ptr(), K, K, None, 0, 0, w.data_ptr(), K, b.data_ptr(),
                                                     None, M, Nn, 1, st())), reps)
        t_dg = t_dg_old = None
        if K % 256 == 0:
            t_dg = timed(lambda: N.check(lib.mnr_tgemm_run(C.byref(d), st())), reps)
            t_dg_old = timed(lambda: N.check(lib.mnr_gemm(dx.data_ptr(), K, y.data_ptr(), Nn, 1, w.data_ptr(), 1, K, M, K, Nn, 0, 1, st())), reps)
        tf = lambda ms: None if ms is None else round(fl / ms / 1e9, 1)
        print(json.dumps({'M': M, 'N': Nn, 'K': K, 'fwd_tiled_tflops': tf(t_new), 'fwd_128x128_tflops': tf(t_old),
                          'dgrad_tiled_tflops': tf(t_dg), 'dgrad_128x128_tflops': tf(t_dg_old),
                          'fwd_ms': round(t_new, 4), 'dgrad_ms': None if t_dg is None else round(t_dg, 4)}), flush=True)


if __name__ == '__main__':
    main()
