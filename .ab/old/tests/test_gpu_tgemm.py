"""Tiled GEMM (csrc/tgemm.hip) and the job form of the batched weight gradients (csrc/wgrad.hip) that serve the wide layers
of the layer-by-layer path, against fp64 torch on the CPU.  Tolerances: exact-fp32 MFMA accumulation, so 2e-6 of the
largest magnitude per output (K <= 640)."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def T(a):
    return torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float32).to(DEV)


def close(got, ref, rel):
    ref = ref.numpy() if isinstance(ref, torch.Tensor) else ref
    err = float(np.abs(got.cpu().numpy().astype(np.float64) - ref).max())
    assert err <= rel * max(float(np.abs(ref).max()), 1e-20), (err, float(np.abs(ref).max()))


@pytest.mark.parametrize('tile_rows', ['256', '128'])
@pytest.mark.parametrize('M', [700, 256, 1])
def test_forward_two_phases_bias_relu(M, tile_rows, monkeypatch):
    from mega_nerf import _native as N
    monkeypatch.setenv('MNR_TGEMM_TILE_ROWS', tile_rows)          # both tile heights (the library picks by row count otherwise)
    rng = np.random.default_rng(3)
    K1, K2, Nn = 64, 256, 512
    x1 = rng.standard_normal((M, K1)).astype(np.float32)
    x2 = rng.standard_normal((M, K2 + 32)).astype(np.float32)          # pitch > K2
    w = (rng.standard_normal((Nn, K1 + K2)) / 8).astype(np.float32)
    b = rng.standard_normal(Nn).astype(np.float32)
    for relu in (1, 0):
        ref = torch.tensor(x1, dtype=torch.float64) @ torch.tensor(w[:, :K1], dtype=torch.float64).T \
            + torch.tensor(x2[:, :K2], dtype=torch.float64) @ torch.tensor(w[:, K1:], dtype=torch.float64).T + torch.tensor(b, dtype=torch.float64)
        if relu:
            ref = torch.relu(ref)
        x1t, x2t, wt, bt = T(x1), T(x2), T(w), T(b)
        y = torch.full((M, Nn + 4), 7.0, device=DEV)
        g = N.TGemm()
        g.a[0], g.lda[0], g.b[0], g.ldb[0], g.k[0] = x1t.data_ptr(), K1, wt.data_ptr(), K1 + K2, K1
        g.a[1], g.lda[1], g.b[1], g.ldb[1], g.k[1] = x2t.data_ptr(), K2 + 32, wt.data_ptr() + 4 * K1, K1 + K2, K2
        g.n_phases, g.b_kslow, g.relu = 2, 0, relu
        g.c, g.ldc, g.m, g.n, g.bias = y.data_ptr(), Nn + 4, M, Nn, bt.data_ptr()
        N.check(N.lib().mnr_tgemm_run(C.byref(g), None))
        torch.cuda.synchronize()
        close(y[:, :Nn], ref, 2e-6)
        assert float(y[:, Nn:].min()) == 7.0                              # columns past n untouched


@pytest.mark.parametrize('tile_rows', ['256', '128'])
@pytest.mark.parametrize('mode', ['plain', 'gate', 'gate_r1'])
def test_data_gradient_forms(mode, tile_rows, monkeypatch):
    from mega_nerf import _native as N
    monkeypatch.setenv('MNR_TGEMM_TILE_ROWS', tile_rows)
    rng = np.random.default_rng(4)
    M, n_out, k_in, col0 = 900, 512, 512, 64
    gz = rng.standard_normal((M, n_out)).astype(np.float32)
    w = (rng.standard_normal((n_out, col0 + k_in)) / 8).astype(np.float32)
    h = rng.standard_normal((M, k_in)).astype(np.float32)
    r1r = rng.standard_normal(M).astype(np.float32)
    r1c = rng.standard_normal(k_in).astype(np.float32)
    ref = torch.tensor(gz, dtype=torch.float64) @ torch.tensor(w[:, col0:], dtype=torch.float64)
    if mode == 'gate_r1':
        ref = ref + torch.tensor(r1r, dtype=torch.float64)[:, None] * torch.tensor(r1c, dtype=torch.float64)[None]
    if mode != 'plain':
        ref = ref * torch.tensor(h > 0, dtype=torch.float64)
    gt, wt, ht, rr, rc = T(gz), T(w), T(h), T(r1r), T(r1c)
    out = torch.empty(M, k_in, device=DEV)
    g = N.TGemm()
    g.a[0], g.lda[0], g.b[0], g.ldb[0], g.k[0] = gt.data_ptr(), n_out, wt.data_ptr() + 4 * col0, col0 + k_in, n_out
    g.n_phases, g.b_kslow = 1, 1
    g.c, g.ldc, g.m, g.n = out.data_ptr(), k_in, M, k_in
    if mode != 'plain':
        g.gate, g.ldgate = ht.data_ptr(), k_in
    if mode == 'gate_r1':
        g.r1_row, g.r1_stride, g.r1_col = rr.data_ptr(), 1, rc.data_ptr()
    N.check(N.lib().mnr_tgemm_run(C.byref(g), None))
    torch.cuda.synchronize()
    close(out, ref, 2e-6)


def test_tgemm_rejects_unsupported_shapes():
    from mega_nerf import _native as N
    x = torch.zeros(64, 64, device=DEV)
    g = N.TGemm()
    g.a[0], g.lda[0], g.b[0], g.ldb[0], g.k[0] = x.data_ptr(), 64, x.data_ptr(), 64, 48       # k not a multiple of 32
    g.n_phases, g.c, g.ldc, g.m, g.n, g.bias = 1, x.data_ptr(), 64, 64, 256, x.data_ptr()
    assert N.lib().mnr_tgemm_run(C.byref(g), None) != 0
    g.k[0], g.n = 64, 100                                                                       # n not a multiple of 256
    assert N.lib().mnr_tgemm_run(C.byref(g), None) != 0


@pytest.mark.parametrize('rows', [640, 4096 + 32])
def test_weight_gradient_jobs(rows):
    """layer_dim 512 skip layer: dW [512][63 + 512] from dZ [rows][512], [embedding (64 dense, 63 used) | hidden [rows][512]]."""
    from mega_nerf import _native as N
    lib = N.lib()
    rng = np.random.default_rng(5)
    W, E = 512, 63
    dz = rng.standard_normal((rows, W)).astype(np.float32)
    hid = rng.standard_normal((rows, W)).astype(np.float32)
    emb = np.zeros((rows, 64), np.float32)
    emb[:, :E] = rng.standard_normal((rows, E))
    emb[:, E:] = 5.0                                                   # padding column must not leak into the gradient
    dzt, ht, et = T(dz), T(hid), T(emb)
    dw = torch.full((W, E + W), 0.5, device=DEV)                       # gradients are accumulated
    db = torch.full((W,), -1.0, device=DEV)
    jobs = []
    for mh in range(2):
        j = N.WgradJob()
        j.dz, j.ldz, j.in_, j.ldin, j.in_cols, j.in_block = dzt.data_ptr() + 4 * 256 * mh, W, et.data_ptr(), 64, E, 64
        j.dw, j.ldw, j.db = dw.data_ptr() + 4 * 256 * mh * (E + W), E + W, db.data_ptr() + 4 * 256 * mh
        jobs.append(j)
        for nh in range(2):
            j = N.WgradJob()
            j.dz, j.ldz, j.in_, j.ldin, j.in_cols, j.in_block = dzt.data_ptr() + 4 * 256 * mh, W, ht.data_ptr() + 4 * 256 * nh, W, 256, 256
            j.dw, j.ldw = dw.data_ptr() + 4 * (256 * mh * (E + W) + E + 256 * nh), E + W
            jobs.append(j)
    arr = (N.WgradJob * len(jobs))(*jobs)
    ws = torch.empty(lib.mnr_wgrad_workspace_bytes(), dtype=torch.uint8, device=DEV)
    N.check(lib.mnr_wgrad_jobs(arr, len(jobs), rows, ws.data_ptr(), ws.numel(), None))
    torch.cuda.synchronize()
    dz64 = torch.tensor(dz, dtype=torch.float64)
    ref = 0.5 + dz64.T @ torch.tensor(np.concatenate([emb[:, :E], hid], 1), dtype=torch.float64)
    close(dw, ref, 3e-6)
    close(db, -1.0 + dz64.sum(0), 3e-6)
    assert lib.mnr_wgrad_jobs(arr, len(jobs), rows + 8, ws.data_ptr(), ws.numel(), None) != 0     # rows must be whole tiles
