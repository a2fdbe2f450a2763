"""Rank-0 logging helpers (reference: mega_nerf/misc_utils.py:6-15)."""
import os


def is_main() -> bool:
    return int(os.environ.get('LOCAL_RANK', 0)) == 0


def main_print(log) -> None:
    if is_main():
        print(log, flush=True)


def main_tqdm(inner):
    if not is_main():
        return inner
    try:
        from tqdm import tqdm
        return tqdm(inner)
    except ImportError:      # pragma: no cover
        return inner
