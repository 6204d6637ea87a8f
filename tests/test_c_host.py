"""Hosts without Python on the C ABI.  tests/c_host/setup_host.c drives the set-up entry
points that run on the library's host threads (no GPU needed: it runs in the CPU suite).
tests/c_host/kron_host.c is plain C99, compiled
with gcc against include/stk.h and linked with libstk.so -- plan from CSR arrays, slab
storage, the Kronecker apply and a dot product, checked inside the program against the
triple loop of the definition.  Without a GPU the program is compiled and linked (the
header is valid C, every symbol it uses resolves); with one it runs."""
import os
import shutil
import subprocess

import pytest

from conftest import PKG, REPO

SRC = os.path.join(REPO, 'tests', 'c_host', 'kron_host.c')
SETUP_SRC = os.path.join(REPO, 'tests', 'c_host', 'setup_host.c')
ROCM = os.environ.get('ROCM_PATH', '/opt/rocm')


def _build(out, src=SRC):
    # a host without the ROCm headers, libamdhip64 or a built libstk.so (a sanitiser
    # build of the host library, a CPU-only machine) cannot compile or link the
    # program: skip, do not fail the CPU suite
    missing = [what for what, path in (('gcc', shutil.which('gcc')),
                                       ('hip_runtime_api.h', os.path.join(ROCM, 'include', 'hip', 'hip_runtime_api.h')),
                                       ('libamdhip64.so', os.path.join(ROCM, 'lib', 'libamdhip64.so')),
                                       ('libstk.so', os.path.join(PKG, 'libstk.so')))
               if not (path and os.path.exists(path))]
    if missing:
        pytest.skip('cannot build the C host here: %s missing' % ', '.join(missing))
    cmd = ['gcc', '-std=c99', '-O2', '-Wall', '-Werror', '-D__HIP_PLATFORM_AMD__',
           '-I' + os.path.join(REPO, 'include'), '-I' + os.path.join(ROCM, 'include'), src, '-o', out,
           '-L' + PKG, '-lstk', '-L' + os.path.join(ROCM, 'lib'), '-lamdhip64', '-lm',
           '-Wl,-rpath,' + PKG, '-Wl,-rpath,' + os.path.join(ROCM, 'lib')]
    res = subprocess.run(cmd, capture_output=True, text=True)
    assert res.returncode == 0, res.stdout + res.stderr
    return out


def test_c_host_compiles_and_links(tmp_path):
    _build(str(tmp_path / 'kron_host'))


@pytest.mark.parametrize('levels', [1, 5, 7])
def test_c_host_runs_the_set_up_calls(tmp_path, levels):
    """tests/c_host/setup_host.c: the set-up entry points that run on the host threads of
    the library -- stk_tri_refine, stk_p1_assemble_2d, stk_p1_load_*, stk_csr_union_*,
    stk_tile_order -- from plain C, on the unit square, checked in the program against
    what the mesh dictates (vertex and entry counts of the three-direction grid, the area,
    the integral of x, the union pattern of (M, A) being M's).  No GPU is touched."""
    exe = _build(str(tmp_path / 'setup_host'), SETUP_SRC)
    res = subprocess.run([exe, str(levels)], capture_output=True, text=True, timeout=300)
    assert res.returncode == 0 and 'setup_host ok' in res.stdout, res.stdout + res.stderr
    side = 2 ** levels
    assert '%d vertices' % ((side + 1) ** 2) in res.stdout and '%d free dofs' % ((side - 1) ** 2) in res.stdout


@pytest.mark.gpu
@pytest.mark.parametrize('shape', [(41, 37, 17), (64, 50, 65), (30, 30, 8), (25, 20, 1)])
def test_c_host_runs_the_kronecker_apply(tmp_path, shape):
    exe = _build(str(tmp_path / 'kron_host'))
    res = subprocess.run([exe] + [str(v) for v in shape], capture_output=True, text=True, timeout=300)
    assert res.returncode == 0 and 'kron_host ok' in res.stdout, res.stdout + res.stderr
    # the grid's matrices have a dictionary: the packed form was built and, from 8 steps
    # on, served row pairs
    assert 'packed=1' in res.stdout
