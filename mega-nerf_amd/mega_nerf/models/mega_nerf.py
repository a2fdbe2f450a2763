"""Spatial router over per-cell NeRFs (reference: mega_nerf/models/mega_nerf.py:7-61).

Routing runs on the device (``k_route``: blend weights of every cell + per-cell row lists appended with
wave-aggregated atomics); each cell then evaluates only its own rows through the fused MLP kernel in gather mode
(``mnr_mlp_io.row_index``) and ``k_route_accumulate`` blends the results in cell order -- the same summation order
as the reference loop, with no host synchronisation (the reference syncs once per cell, mega_nerf.py:38).
"""
import ctypes as C
from typing import List, Optional

import torch
from torch import nn

from mega_nerf import _native as N


def _arch_key(m):
    return (m.xyz_dim, m.pos_xyz_dim, m.pos_dir_dim, m.layers, tuple(m.skip_layers), m.layer_dim, m.appearance_dim,
            m.appearance_count, m.rgb_dim, type(m.sigma_activation).__name__, m.mfma_tile)


class RoutedTape:
    """Tapes of one routed training evaluation: per cell (parameter prefix, cell tape, routed rows, blend weights)."""

    def __init__(self):
        self.cells = []

    def backward(self, d_out: torch.Tensor, d_out_stride: int, grads: dict) -> None:
        d2 = d_out.view(-1, d_out_stride)
        for prefix, cell_tape, rows, w, _keepalive in self.cells:
            d_sub = d2.index_select(0, rows)
            if w is not None:
                d_sub = d_sub * w[:, None]
            cell_tape.backward(d_sub.contiguous(), d_sub.shape[1], {k[len(prefix):]: v for k, v in grads.items() if k.startswith(prefix)})


class _RoutedJob:
    """State of one routed evaluation between MegaNeRF._route_begin and ._route_finish."""
    __slots__ = ('args', 'out', 'B', 'ncol', 'n_units', 'rows_per_unit', 'weights', 'lists', 'counts', 'inverse', 'cells', 'split', 'sub_out',
                 'desc', 'io', 'default_arch')


def evaluate_routed_together(reqs) -> bool:
    """The routed evaluations of several containers (the foreground and the background container of one render pass) as ONE
    gather-mode launch (mnr_mlp_forward_cells_multi): ``reqs`` = [(container, xyz, part, S, out, noise, sh_deg)], as
    :meth:`MegaNeRF.evaluate_routed` takes them.  Returns False -- nothing done -- when a container is not a set of default 8x256
    cells (the caller then evaluates them one by one)."""
    for m, *_ in reqs:
        if not isinstance(m, MegaNeRF):
            return False
        kids = list(m.sub_modules)
        if not (len(kids) <= 64 and all(c.fused_supported() and c.is_default_arch() for c in kids)):
            return False
    from mega_nerf import rendering as R
    if R.SPLIT_PRECISION and not torch.is_grad_enabled():
        return False
    jobs = [m._begin_from_parts(xyz, part, S, out, noise, sh_deg) for m, xyz, part, S, out, noise, sh_deg in reqs]
    if all(j.cells is not None and j.default_arch for j in jobs):
        # smallest segment first: its workgroups start at once and the larger one's fill the chip behind them
        order = sorted(range(len(jobs)), key=lambda i: jobs[i].B)
        segs = (N.MlpCellsLaunch * len(jobs))()
        for sg, i in zip(segs, order):
            sg.desc, sg.cells_dev, sg.n_cells, sg.io = C.pointer(jobs[i].desc), jobs[i].cells.data_ptr(), len(reqs[i][0].sub_modules), C.pointer(jobs[i].io)
        N.check(N.lib().mnr_mlp_forward_cells_multi(segs, len(jobs), N.stream_ptr()))
        for (m, *_), j in zip(reqs, jobs):
            m._route_finish(j)
        return True
    for (m, *_), j in zip(reqs, jobs):         # routed already: finish every job on its own
        if j.cells is not None:
            fwd = N.lib().mnr_mlp_forward_cells_h2 if j.split else N.lib().mnr_mlp_forward_cells
            N.check(fwd(C.byref(j.desc), j.cells.data_ptr(), len(m.sub_modules), C.byref(j.io), N.stream_ptr()))
            m._route_finish(j)
        else:
            m._routed_cell_by_cell(j)
    return True


class MegaNeRF(nn.Module):
    def __init__(self, sub_modules: List[nn.Module], centroids: torch.Tensor, boundary_margin: float, xyz_real: bool,
                 cluster_2d: bool, joint_training: bool = False):
        super().__init__()
        assert boundary_margin >= 1
        self.sub_modules = nn.ModuleList(sub_modules)
        self.register_buffer('centroids', centroids)
        self.boundary_margin = boundary_margin
        self.xyz_real = xyz_real
        self.cluster_dim_start = 1 if cluster_2d else 0
        self.joint_training = joint_training
        self._cent_host = None

    # attributes rendering.py reads from a plain NeRF
    @property
    def has_dir(self):
        return self.sub_modules[0].has_dir

    @property
    def embedding_a(self):
        return self.sub_modules[0].embedding_a

    def _centroids_host(self):
        """Host copy of the centroid buffer for mnr_route, re-read whenever the buffer was replaced or written
        (``load_state_dict`` copies in place: version bump; ``.to()`` swaps the storage: pointer change)."""
        cen = self.centroids
        key = (cen.data_ptr(), -1 if cen.is_inference() else cen._version)
        if self._cent_host is None or self._cent_host[0] != key:
            c = cen.detach().float().cpu().contiguous().view(-1).tolist()
            self._cent_host = (key, (C.c_float * len(c))(*c))
        return self._cent_host[1]

    def _cell_table(self, rows: List[List[int]], dev: torch.device) -> torch.Tensor:
        """Device copy of a launch's ``mnr_mlp_cell`` table.  A pageable host -> device copy blocks the host until the stream has
        drained (four times per render: the routed evaluation then runs at the pace of the host); in steady state the caching
        allocator hands every evaluation the buffers of the previous one, so the last tables are kept (keyed by their content) and
        a changed table goes up from pinned memory without waiting.  A new table is a new tensor: launches still reading the old
        one keep it alive."""
        # keyed by the STREAM as well: the upload is ordered on the stream that was current at first use, and a cache hit on another
        # stream (rendering._render_ws renders per (device, stream)) would launch against the table with nothing ordering it behind
        # that copy
        key = (dev.index if dev.index is not None else torch.cuda.current_device(), torch.cuda.current_stream(dev).cuda_stream) + tuple(v for r in rows for v in r)
        cache = self.__dict__.setdefault('_cell_tables', {})
        hit = cache.get(key)
        if hit is not None:
            return hit
        host = torch.tensor(rows, dtype=torch.int64).pin_memory()
        table = host.to(dev, non_blocking=True)
        if len(cache) >= 16:
            cache.pop(next(iter(cache)))
        cache[key] = table
        return table

    def _route_ws(self, B: int, ncol: int, dev: torch.device) -> dict:
        """The buffers of a routed evaluation of B rows (blend weights, row lists, counts, inverse map, the cells' compact outputs), kept
        per (stream, B, ncol): a render evaluates the same row counts every time, and fresh ``torch.empty`` buffers each time moved the
        pointers of the launch's cell table -- a table upload per evaluation.  One evaluation of a given size is in flight per container
        and stream (the next one's kernels are ordered behind the previous one's on that stream); at most four sizes are kept."""
        cache = self.__dict__.setdefault('_route_buffers', {})
        # (the device is part of the key: the default stream's handle is 0 on every GPU, so after .to(other_device) a hit would hand
        # back buffers of the old device)
        key = (dev.index if dev.index is not None else torch.cuda.current_device(), torch.cuda.current_stream(dev).cuda_stream, B, ncol)
        ws = cache.get(key)
        if ws is None:
            n_sub = len(self.sub_modules)
            ws = dict(weights=torch.empty(n_sub, B, device=dev, dtype=torch.float32), lists=torch.empty(n_sub, B, device=dev, dtype=torch.int32),
                      counts=torch.empty(n_sub, device=dev, dtype=torch.int32), inverse=torch.empty(n_sub, B, device=dev, dtype=torch.int32),
                      sub_out=None)
            if len(cache) >= 4:
                cache.pop(next(iter(cache)))
            cache[key] = ws
        return ws

    def release_buffers(self) -> None:
        """Drop the cached routing buffers and cell tables (n_sub x rows x 28 bytes + the cells' outputs per kept size: ~9 GB for a
        25-cell container after 65 536-ray x 192-sample evaluations); the next evaluation re-allocates."""
        self.__dict__.pop('_route_buffers', None)
        self.__dict__.pop('_cell_tables', None)

    def _routed(self, pos: torch.Tensor, pos_stride: int, xyz: torch.Tensor, xyz_stride: int,
                dirs: Optional[torch.Tensor], dir_stride: int, idx: Optional[torch.Tensor], idx_stride: int,
                rows_per_ray: int, B: int, out: torch.Tensor, noise: Optional[torch.Tensor], sigma_only: bool, sh_deg: int,
                n_units: Optional[torch.Tensor], rows_per_unit: int) -> None:
        """out [B, C] = sum_i w_i * cell_i(rows routed to i).  All pointers are views into caller-owned buffers."""
        lib = N.lib()
        job = self._route_begin(pos, pos_stride, xyz, xyz_stride, dirs, dir_stride, idx, idx_stride, rows_per_ray, B, out, noise,
                                sigma_only, sh_deg, n_units, rows_per_unit)
        if job.cells is not None:
            # one launch for all cells: each cell alone (~1/n of the rows) cannot fill 256 CUs
            fwd = lib.mnr_mlp_forward_cells_h2 if job.split else lib.mnr_mlp_forward_cells
            N.check(fwd(C.byref(job.desc), job.cells.data_ptr(), len(self.sub_modules), C.byref(job.io), N.stream_ptr()))
            self._route_finish(job)
            return
        self._routed_cell_by_cell(job)

    def _route_begin(self, pos, pos_stride, xyz, xyz_stride, dirs, dir_stride, idx, idx_stride, rows_per_ray, B, out, noise,
                     sigma_only, sh_deg, n_units, rows_per_unit) -> '_RoutedJob':
        """Routing of one evaluation (k_route: blend weights + per-cell row lists) and, when all cells share a fused architecture, the
        argument block of their ONE gather-mode launch (``job.cells`` / ``job.io``; ``None``: cell-by-cell fallback).  The caller
        launches -- alone (:meth:`_routed`) or together with another container's job (:func:`evaluate_routed_together`) -- and
        then calls :meth:`_route_finish`."""
        lib = N.lib()
        dev = out.device
        n_sub = len(self.sub_modules)
        ncol = out.shape[1]
        job = _RoutedJob()
        job.cells = job.sub_out = job.desc = job.io = None
        job.split = job.default_arch = False
        job.args = (xyz, xyz_stride, dirs, dir_stride, idx, idx_stride, rows_per_ray, B, noise, sigma_only, sh_deg)
        job.out, job.B, job.ncol, job.n_units, job.rows_per_unit = out, B, ncol, n_units, rows_per_unit
        ws = self._route_ws(B, ncol, dev)
        job.weights, job.lists, job.counts, job.inverse = weights, lists, counts, inverse = ws['weights'], ws['lists'], ws['counts'], ws['inverse']
        N.check(lib.mnr_route_indexed(pos.data_ptr(), pos_stride, B, N.ptr(n_units), rows_per_unit, self._centroids_host(), n_sub,
                                      self.cluster_dim_start, float(self.boundary_margin), weights.data_ptr(), lists.data_ptr(),
                                      counts.data_ptr(), inverse.data_ptr(), N.stream_ptr()))
        rr = getattr(self, 'routed_rows', None)        # optional device-side tally of routed rows (bench.py: FLOPs of a routed step)
        if rr is not None:
            rr.add_(counts.sum())
        kids = list(self.sub_modules)
        # (mnr_mlp_forward_cells takes at most 64 cells per launch: larger grids go cell by cell)
        same_arch = n_sub <= 64 and all(c.fused_supported() and _arch_key(c) == _arch_key(kids[0]) for c in kids)
        job.cells = None
        if not same_arch:
            out.zero_()                                # (the cell-by-cell fallback accumulates into `out`)
            return job
        from mega_nerf import rendering as R
        # opt-in split precision (rendering.SPLIT_PRECISION; csrc/mlp_fwd_h2.hip): inference of the default 8x256 cells
        split = (R.SPLIT_PRECISION and not torch.is_grad_enabled() and not sigma_only and sh_deg < 0 and dirs is not None and
                 idx is not None and all(c.is_default_arch() for c in kids))
        self._last_routed_split = job.split = split       # (tests: which kernel family served the last routed evaluation)
        if ws['sub_out'] is None:
            ws['sub_out'] = torch.empty(n_sub, B, ncol, device=dev, dtype=torch.float32)
        job.sub_out = sub_out = ws['sub_out']
        rows = []
        for i, child in enumerate(kids):
            _, packed = child.packed_h2() if split else child.packed()
            rows.append([packed.data_ptr(), child.embedding_a.weight.data_ptr() if child.embedding_a is not None else 0,
                         lists[i].data_ptr(), counts[i:i + 1].data_ptr(), sub_out[i].data_ptr()])
        job.cells = self._cell_table(rows, dev)                               # mnr_mlp_cell[n_sub]
        job.desc, _ = kids[0].packed()
        job.io = io = kids[0].mlp_io(xyz, xyz_stride, dirs, dir_stride, idx, idx_stride, rows_per_ray, B, sub_out[0], noise, None, 0)
        io.sigma_only = 1 if sigma_only else 0
        io.apply_sh_deg = sh_deg
        # the default architectures, inference of all four outputs: the job can share a launch with another container's
        job.default_arch = (not split and not sigma_only and sh_deg < 0 and dirs is not None and idx is not None and
                            all(c.is_default_arch() for c in kids))
        return job

    def _route_finish(self, job: '_RoutedJob') -> None:
        """Blend the cells' outputs in cell order (k_route_combine: the reference loop's summation order, mega_nerf.py:28-49)."""
        n_sub = len(self.sub_modules)
        # (the inverse map came out of the routing kernel: one launch, and rows past the device-side count are zeroed by it)
        N.check(N.lib().mnr_route_combine_indexed(job.out.data_ptr(), job.ncol, job.sub_out.data_ptr(), job.B * job.ncol, job.ncol, job.ncol,
                                                  job.inverse.data_ptr(), job.weights.data_ptr() if self.boundary_margin > 1 else None,
                                                  n_sub, job.B, N.ptr(job.n_units), job.rows_per_unit, N.stream_ptr()))

    def _routed_cell_by_cell(self, job: '_RoutedJob') -> None:
        lib = N.lib()
        xyz, xyz_stride, dirs, dir_stride, idx, idx_stride, rows_per_ray, B, noise, sigma_only, sh_deg = job.args
        out, ncol, lists, counts, weights = job.out, job.ncol, job.lists, job.counts, job.weights
        dev = out.device
        blend = self.boundary_margin > 1
        sub_out = torch.empty(B, ncol, device=dev, dtype=torch.float32)
        for i, child in enumerate(self.sub_modules):
            if not child.fused_supported():
                self._child_gathered(child, lists[i], counts[i:i + 1], xyz, xyz_stride, dirs, dir_stride, idx, idx_stride,
                                     rows_per_ray, B, sub_out, noise, sigma_only, sh_deg)
                N.check(lib.mnr_route_accumulate(out.data_ptr(), ncol, sub_out.data_ptr(), ncol, ncol, lists[i].data_ptr(),
                                                 counts[i:i + 1].data_ptr(), B, weights[i].data_ptr() if blend else None,
                                                 0 if blend else 1, N.stream_ptr()))
                continue
            io = child.mlp_io(xyz, xyz_stride, dirs, dir_stride, idx, idx_stride, rows_per_ray, B, sub_out, noise,
                              counts[i:i + 1], 1)
            io.row_index = lists[i].data_ptr()
            io.sigma_only = 1 if sigma_only else 0
            io.apply_sh_deg = sh_deg
            child.launch(io)
            N.check(lib.mnr_route_accumulate(out.data_ptr(), ncol, sub_out.data_ptr(), ncol, ncol, lists[i].data_ptr(),
                                             counts[i:i + 1].data_ptr(), B, weights[i].data_ptr() if blend else None,
                                             0 if blend else 1, N.stream_ptr()))

    @staticmethod
    def _child_gathered(child, rows_list, count, xyz, xyz_stride, dirs, dir_stride, idx, idx_stride, rows_per_ray, B, sub_out,
                        noise, sigma_only, sh_deg) -> None:
        """Cells without a fused kernel (generic widths): materialise the routed rows and run the layer-by-layer path.
        Sizes its launches on the host (one device->host read of the row count per cell)."""
        cnt = int(count.item())
        if cnt == 0:
            return
        rows = rows_list[:cnt].long()
        rays = rows // rows_per_ray
        n_rays = (B + rows_per_ray - 1) // rows_per_ray
        x2 = torch.as_strided(xyz, (B, child.xyz_dim), (xyz_stride, 1), xyz.storage_offset()).index_select(0, rows)
        d2 = i2 = None
        if dirs is not None:
            d2 = torch.as_strided(dirs, (n_rays, 3), (dir_stride, 1), dirs.storage_offset()).index_select(0, rays)
        if idx is not None:
            i2 = torch.as_strided(idx, (n_rays,), (idx_stride,), idx.storage_offset()).index_select(0, rays)
        n2 = noise.index_select(0, rows) if noise is not None else None
        child.evaluate(x2, child.xyz_dim, d2, 3, i2, 1, 1, cnt, sub_out, n2, sigma_only, sh_deg)

    def train_eval_routed(self, xyz: torch.Tensor, part, S: int, out: torch.Tensor, noise, sh_deg: int) -> 'RoutedTape':
        """Training-mode twin of :meth:`evaluate_routed` (``--train_mega_nerf``: all cells trained in one process,
        mega_nerf.py:28-59): rows are routed on the device, every cell evaluates its gathered rows with its own tape
        (``NeRF.train_eval``), and the tape of the whole evaluation replays the cells backwards.  Sizes its per-cell
        launches on the host (one read of the row counts per evaluation)."""
        lib = N.lib()
        dev = out.device
        n, ncol_in = xyz.shape[0], xyz.shape[-1]
        B, n_sub = n * S, len(self.sub_modules)
        if part.n_units is not None:
            B = min(B, int(part.n_units.item()) * S)
        flat = xyz.view(-1, ncol_in)
        out2 = out.view(-1, out.shape[-1])
        tape = RoutedTape()
        if B == 0:
            return tape
        weights = torch.empty(n_sub, B, device=dev, dtype=torch.float32)
        lists = torch.empty(n_sub, B, device=dev, dtype=torch.int32)
        counts = torch.empty(n_sub, device=dev, dtype=torch.int32)
        N.check(lib.mnr_route(flat.data_ptr(), ncol_in, B, None, 0, self._centroids_host(), n_sub, self.cluster_dim_start,
                              float(self.boundary_margin), weights.data_ptr(), lists.data_ptr(), counts.data_ptr(), N.stream_ptr()))
        host_counts = counts.cpu().tolist()
        out2[:B].zero_()
        blend = self.boundary_margin > 1
        x_in = flat[:, 3:] if self.xyz_real else flat
        for i, child in enumerate(self.sub_modules):
            cnt = host_counts[i]
            if cnt == 0:
                continue
            rows = lists[i, :cnt].long()
            rays = rows // S
            xi = x_in.index_select(0, rows).contiguous()
            dirs_i = idx_i = None
            if child.has_dir and child.embedding_a is None:              # quirk Q8 (nerf.py:146)
                dirs_i = torch.cat([xi[:, -1:], part.dirs.index_select(0, rays)[:, :2]], -1).contiguous()
            elif child.has_dir or sh_deg >= 0:
                dirs_i = part.dirs.index_select(0, rays).contiguous()
            if child.embedding_a is not None:
                idx_i = part.idx.index_select(0, rays).contiguous()
            noise_i = noise.index_select(0, rows) if noise is not None else None
            sub = torch.empty(cnt, out2.shape[1], device=dev, dtype=torch.float32)
            cell_tape = child.train_eval(xi, xi.shape[1], dirs_i if child.has_dir else None, 3, 1, idx_i, 1, 1, cnt, sub, noise_i,
                                         sh_deg, None, 0, dirs_i if sh_deg >= 0 else None, 3)
            w = weights[i].index_select(0, rows) if blend else None
            out2.index_add_(0, rows, sub * w[:, None] if blend else sub)
            tape.cells.append(('sub_modules.%d.' % i, cell_tape, rows, w, (xi, dirs_i, idx_i, noise_i, sub)))
        return tape

    def _routed_args(self, xyz: torch.Tensor, part, S: int, out: torch.Tensor, noise, sh_deg: int) -> tuple:
        n, ncol_in = xyz.shape[0], xyz.shape[-1]
        child0 = self.sub_modules[0]
        need_dir = child0.has_dir or sh_deg >= 0
        x_in = xyz.view(-1, ncol_in)[:, 3:] if self.xyz_real else xyz.view(-1, ncol_in)   # pointer offset only
        return (xyz, ncol_in, x_in, ncol_in, part.dirs if need_dir else None, part.dirs.stride(0) if need_dir else 0,
                part.idx if child0.embedding_a is not None else None, 1, S, n * S, out.view(-1, out.shape[-1]), noise,
                False, sh_deg, part.n_units, S)

    def _begin_from_parts(self, xyz, part, S, out, noise, sh_deg) -> '_RoutedJob':
        return self._route_begin(*self._routed_args(xyz, part, S, out, noise, sh_deg))

    def evaluate_routed(self, xyz: torch.Tensor, part, S: int, out: torch.Tensor, noise, sh_deg: int):
        """Render-path entry: xyz [n, S, 3] (fg) or [n, S, 7] = [xyz_real | sphere point | 1/r] (bg, quirk Q15);
        per-ray dirs / image indices in ``part``; ``out`` [n, S, 4]."""
        if sh_deg >= 0 and self.boundary_margin > 1 and self.sub_modules[0].rgb_dim > 3:
            # Spherical-harmonics cells under a soft blend: the reference blends the cells' RAW outputs -- the SH coefficients and sigma
            # (mega_nerf.py:45-49) -- and evaluates eval_sh + sigmoid on the blend (rendering.py:300-306); a blend of the cells' colours
            # AFTER their sigmoids is a different number wherever two cells meet (round 6: the fixture render_container_sh2_eval found
            # the in-kernel colour epilogue being applied per cell).  So: cells write coefficients, the blend runs over rgb_dim + 1
            # columns, one mnr_sh_apply turns the blended rows into colours.  (Hard routing has one cell per row: the epilogue stays fused.)
            n = xyz.shape[0]
            ncol = self.sub_modules[0].rgb_dim + 1
            raw = torch.empty(n, S, ncol, device=out.device, dtype=torch.float32)
            self._routed(*self._routed_args(xyz, part, S, raw, noise, -1))
            N.check(N.lib().mnr_sh_apply(out.data_ptr(), out.shape[-1], raw.data_ptr(), ncol, part.dirs.data_ptr(), part.dirs.stride(0), S,
                                         sh_deg, n * S, N.stream_ptr()))
            return
        self._routed(*self._routed_args(xyz, part, S, out, noise, sh_deg))

    def forward(self, x: torch.Tensor, sigma_only: bool = False,
                sigma_noise: Optional[torch.Tensor] = None) -> torch.Tensor:
        N.require_device(x, 'x')
        child0 = self.sub_modules[0]
        x = x.contiguous().float()
        B, ncol = x.shape
        off = 3 if self.xyz_real else 0
        x_in = x[:, off:]
        dirs = idx = None
        if not sigma_only:
            if child0.has_dir:
                dirs = x[:, ncol - 4:]
            if child0.embedding_a is not None:
                idx = x[:, ncol - 1:]
        out = torch.empty(B, 1 if sigma_only else child0.rgb_dim + 1, device=x.device, dtype=torch.float32)
        if B == 0:
            return out
        noise = sigma_noise.contiguous().float().view(-1) if sigma_noise is not None else None
        self._routed(x, ncol, x_in, ncol, dirs, ncol, idx, ncol, 1, B, out, noise, sigma_only, -1, None, 0)
        return out
