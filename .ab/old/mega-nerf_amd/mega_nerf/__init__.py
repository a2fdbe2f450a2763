"""MI355X-native drop-in for the hot path of cmusatyalab/mega-nerf.

Same import paths as the reference for the path that was rebuilt:
``mega_nerf.ray_utils`` (get_ray_directions / get_rays / get_rays_batch), ``mega_nerf.rendering.render_rays``,
``mega_nerf.models.*`` (NeRF / Cascade / MegaNeRF / get_nerf / get_bg_nerf).  Everything numerical runs in
``libmeganerf_hip.so`` (hand-written HIP for gfx950) through the C ABI of ``include/mnr_api.h``.
"""
__version__ = '0.1.0'
