import os
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
for p in (ROOT, ROOT / 'mega-nerf_amd', ROOT / 'tests', ROOT / 'tests' / 'golden'):
    if str(p) not in sys.path:
        sys.path.insert(0, str(p))
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')
    # the merged-container format IS a TorchScript archive (reference merge_submodules.py:79); torch 2.10 deprecates the API
    config.addinivalue_line('filterwarnings', 'ignore:.*torch.jit.*is deprecated.*:DeprecationWarning')


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason='no GPU visible')
    for item in items:
        if 'gpu' in item.keywords:
            item.add_marker(skip)


def free_port() -> str:
    """A TCP port nothing listens on right now (rendezvous of the multi-process tests: a fixed port can still be held by the previous
    test's agent for a moment)."""
    import socket
    with socket.socket() as so:
        so.bind(('127.0.0.1', 0))
        return str(so.getsockname()[1])


def loopback_env(env: dict) -> dict:
    """gloo picks its interface from the host name, which need not resolve in a container: pin it to the loopback device when there is one."""
    import os
    if os.path.isdir('/sys/class/net/lo'):
        env.setdefault('GLOO_SOCKET_IFNAME', 'lo')
    return env
