"""The seeded synthetic scene / weight generator lives in ``mega-nerf_amd/synthetic_scene.py`` (bench.py uses it too);
this module re-exports it under the name the fixtures and their generator scripts have always imported."""
import sys
from pathlib import Path

_PKG = str(Path(__file__).resolve().parents[2] / 'mega-nerf_amd')
if _PKG not in sys.path:
    sys.path.append(_PKG)       # appended: the golden generators put the *reference* checkout first (its package is also ``mega_nerf``)

from synthetic_scene import *           # noqa: E402,F401,F403
from synthetic_scene import SCENE, f32, make_weights, model_cfg, pick_rays   # noqa: E402,F401
