"""The interval primitives behind the per-batch tape pruning and the cell-group culling
(sdf_amd/csrc/sdf_interval.h) are __host__ __device__: this builds them into a CPU program
(tests/native/interval_host.hip, host side only) that checks, on random boxes, that every value
computed at a point of the box lies inside the interval computed for the box -- circular_array
(hypot / atan2 / floored modulo / sin / cos), repeat (rint / clip), the monotone easing curves and
the interval product."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')


@pytest.mark.skipif(not os.path.exists(HIPCC), reason='hipcc not installed')
def test_interval_primitives_enclose_sampled_points(tmp_path):
    exe = str(tmp_path / 'interval_host')
    subprocess.check_call([HIPCC, '--offload-host-only', '-O1', '-std=c++17', '-ffp-contract=off', '-w',
                           '-I', os.path.join(ROOT, 'sdf_amd', 'csrc'), '-o', exe,
                           os.path.join(ROOT, 'tests', 'native', 'interval_host.hip')])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout[-4000:]
    assert 'fails 0' in out.stdout
