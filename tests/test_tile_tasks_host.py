"""The tile's task map and k_cull's body on the host (tests/native/tile_tasks_host.py: the device source of `TileTasks` and
`cull_tasks` with the device idioms replaced, a workgroup played by host threads) against a NumPy restatement of their
rules.  No GPU.
* `TileTasks`: every sample of a tile -- the full 33^3 one (512 cubes of 4^3 samples + its three far faces) and ragged ones
  (runs of 64 consecutive samples) -- belongs to exactly one (task, lane);
(`cull_tasks` with its three interval levels and its unit list: tests/test_cull_host.py.)"""
import ctypes
import os
import sys

import numpy as np
import pytest

import fixtures
from conftest import GOLDEN, ROOT

sys.path.insert(0, os.path.join(ROOT, 'tests', 'native'))
import tile_tasks_host
import test_interval_host as tih


@pytest.fixture(scope='module')
def libs(tmp_path_factory):
    if not os.path.exists(tile_tasks_host.HIPCC):
        pytest.skip('hipcc not installed')
    return tile_tasks_host.build(str(tmp_path_factory.mktemp('tile'))), tih._tape_lib(tmp_path_factory)


def _task_samples(lib, n):
    """task -> list of (ix, iy, iz) by the device's map; asserts that idle lanes still return valid indices"""
    out = (ctypes.c_int * 3)()
    tasks = []
    for task in range(lib.tile_ntask(*n)):
        own = []
        for lane in range(64):
            ok = lib.tile_sample(n[0], n[1], n[2], task, lane, out)
            assert 0 <= out[0] < n[0] and 0 <= out[1] < n[1] and 0 <= out[2] < n[2]
            if ok:
                own.append((out[0], out[1], out[2]))
        tasks.append(own)
    return tasks


@pytest.mark.parametrize('shape', [(33, 33, 33), (32, 33, 33), (33, 33, 32), (33, 17, 5), (2, 2, 2), (7, 33, 33), (33, 2, 33), (19, 19, 19), (1, 33, 33)],
                         ids=lambda s: 'x'.join(map(str, s)))
def test_every_sample_of_a_tile_belongs_to_exactly_one_task_lane(shape, libs):
    lib, _ = libs
    seen = np.zeros(shape, np.int32)
    for own in _task_samples(lib, shape):
        for q in own:
            seen[q] += 1
    assert (seen == 1).all()
    if shape == (33, 33, 33):
        assert lib.tile_ntask(*shape) == 563
    else:
        assert lib.tile_ntask(*shape) == (int(np.prod(shape)) + 63) // 64
