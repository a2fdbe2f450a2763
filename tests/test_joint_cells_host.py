"""training.JointCells on the host: the rendezvous of several cells' training loops (one thread each, exactly one running at a time)
without a GPU -- the joint step itself is replaced by a recorder."""
import sys
import threading
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
for p in (ROOT, ROOT / 'mega-nerf_amd'):
    if str(p) not in sys.path:
        sys.path.insert(0, str(p))


class _Batch:
    def __init__(self, tag):
        self.tag = tag


def _group(n, log, fail_at=None):
    from mega_nerf.training import JointCells

    class Recorder(JointCells):
        def _step_all(self):
            assert len(self.pending) == self.n
            tags = [self.pending[i].tag for i in range(self.n)]
            log.append(tags)
            if fail_at is not None and len(log) == fail_at:
                raise ValueError('step %d failed' % fail_at)
            for i in range(self.n):
                self.results[i] = ('loss', i, tags[i])
    return Recorder(n)


def test_loops_meet_once_per_iteration_and_never_run_concurrently():
    log, running, overlaps = [], [0], [0]
    g = _group(3, log)

    def loop(i):
        def body():
            for it in range(20):
                running[0] += 1
                if running[0] > 1:
                    overlaps[0] += 1
                # (work between two steps: under the baton, so no other loop may be inside it)
                x = sum(range(200))
                running[0] -= 1
                res = g.submit(i, _Batch((i, it, x)))
                assert res == ('loss', i, (i, it, x))
        return body
    g.run([loop(i) for i in range(3)])
    assert overlaps[0] == 0
    assert len(log) == 20 and all([t[1] for t in tags] == [it] * 3 and [t[0] for t in tags] == [0, 1, 2] for it, tags in enumerate(log))


def test_an_exception_in_one_loop_or_in_the_joint_step_ends_all_of_them():
    log = []
    g = _group(2, log)

    def good():
        for it in range(10):
            g.submit(0, _Batch(it))

    def bad():
        for it in range(10):
            if it == 3:
                raise KeyError('cell 1 broke')
            g.submit(1, _Batch(it))
    with pytest.raises((KeyError, RuntimeError)):
        g.run([good, bad])
    assert len(log) == 3
    log2 = []
    g2 = _group(2, log2, fail_at=4)
    with pytest.raises((ValueError, RuntimeError)):
        g2.run([lambda: [g2.submit(0, _Batch(i)) for i in range(10)], lambda: [g2.submit(1, _Batch(i)) for i in range(10)]])
    assert len(log2) == 4
    assert threading.active_count() < 10


def test_a_loop_that_ends_early_does_not_leave_the_others_waiting():
    log = []
    g = _group(2, log)
    with pytest.raises(RuntimeError):
        g.run([lambda: [g.submit(0, _Batch(i)) for i in range(5)], lambda: [g.submit(1, _Batch(i)) for i in range(3)]])
    assert len(log) == 3
