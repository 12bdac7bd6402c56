import os
import sys
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')
    # The shared libraries are build artefacts (git-ignored; they travel to the GPU box inside the repo snapshot).  In a
    # fresh checkout build them on demand — hipcc cross-compiles gfx950 without a GPU.  On a box without compilers the
    # imports below fail loudly, as the product itself does.
    import shutil
    from pcgcv2_amd import _lib
    if not os.path.exists(_lib.LIB_PATH) and (shutil.which('hipcc') or os.path.exists('/opt/rocm/bin/hipcc')):
        from pcgcv2_amd import _build
        _build.build()
    from oracle import pcgc_oracle
    if not os.path.exists(pcgc_oracle._SO) and shutil.which('gcc'):
        pcgc_oracle.build()


@pytest.fixture(scope='session')
def golden_dir():
    return GOLDEN


@pytest.fixture(scope='session', autouse=True)
def _oracle_threads():
    """The oracle's OpenMP loops use the CPUs this container may really use (cgroup quota), not every advertised core."""
    import pcgcv2_amd
    from oracle import pcgc_oracle
    pcgc_oracle.set_threads(pcgcv2_amd.effective_cpus())
    yield
