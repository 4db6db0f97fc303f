"""The drop-in boundary is a C ABI: examples/lstur_step.c (plain C, gcc, no Python / torch types) must compile
against include/lstur.h and link liblstur_hip.so; on an MI355X it trains 20 steps and the loss falls."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIBDIR = os.path.join(ROOT, 'mnexp_amd')


def build(out):
    cmd = ['gcc', '-O2', '-Wall', '-D__HIP_PLATFORM_AMD__', '-I/opt/rocm/include', '-I' + os.path.join(ROOT, 'include'),
           os.path.join(ROOT, 'examples', 'lstur_step.c'), '-L' + LIBDIR, '-llstur_hip', '-L/opt/rocm/lib', '-lamdhip64',
           '-Wl,-rpath,' + LIBDIR, '-Wl,-rpath,/opt/rocm/lib', '-lm', '-o', out]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return out


def test_c_example_compiles_and_links(tmp_path):
    if not os.path.exists(os.path.join(LIBDIR, 'liblstur_hip.so')):
        pytest.skip('library not built')
    exe = build(str(tmp_path / 'lstur_step'))
    assert os.path.getsize(exe) > 0


@pytest.mark.gpu
def test_c_example_trains(tmp_path):
    torch = pytest.importorskip('torch')
    if not torch.cuda.is_available():
        pytest.skip('needs an MI355X')
    exe = build(str(tmp_path / 'lstur_step'))
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert 'ok: loss' in r.stdout
