"""MI355X-native drop-in for the hot path of cmusatyalab/mega-nerf.

Same import paths as the reference for the path that was rebuilt:
``mega_nerf.ray_utils`` (get_ray_directions / get_rays / get_rays_batch), ``mega_nerf.rendering.render_rays``,
``mega_nerf.models.*`` (NeRF / Cascade / MegaNeRF / get_nerf / get_bg_nerf).  Everything numerical runs in
``libmeganerf_hip.so`` (hand-written HIP for gfx950) through the C ABI of ``include/mnr_api.h``.
"""
import os as _os

# Kernel arguments in device memory (the default of this ROCm release; the HIP runtime reads the variable when it initialises, i.e. at the
# first device call of the process): with host-resident arguments every wavefront's argument reads cross the host link -- the ray-stage
# kernels measured +4 ... +27 % on a healthy box, more where that link is slow.  An explicit setting of the caller wins.
_os.environ.setdefault('HIP_FORCE_DEV_KERNARG', '1')

__version__ = '0.1.0'
