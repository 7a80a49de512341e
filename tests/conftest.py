import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
for p in (str(ROOT), str(ROOT / "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")
    # The in-tree binaries are git-ignored build products; if a checkout arrives without them, build them once (hipcc
    # cross-compiles gfx950 without a GPU).  Nothing is built when they are present.
    need = [ROOT / "opengl-raytracing_amd" / "librt_mi355.so", ROOT / "opengl-raytracing_amd" / "rt_cli", ROOT / "oracle" / "liborc.so"]
    if not all(f.exists() for f in need):
        import __graft_entry__
        __graft_entry__.build()


@pytest.fixture(scope="session")
def orc():
    import oracle
    oracle.lib()
    return oracle
