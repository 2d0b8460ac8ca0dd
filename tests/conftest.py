import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


def pytest_terminal_summary(terminalreporter, exitstatus, config):
    """the full-size parity tests skip when the host cannot hold the oracle's f32 weights (48 GB, 96 GB with the 8-bit recipes'
    images).  A skip keeps rc 0, so a smaller box would turn the headline parity rows into silence: say it LOUDLY at the end of the run."""
    skipped = terminalreporter.stats.get("skipped", [])
    big = [r for r in skipped if "host cannot hold" in str(getattr(r, "longrepr", ""))]
    multi = [r for r in skipped if "device" in str(getattr(r, "longrepr", "")).lower() and r not in big]
    tr = terminalreporter
    if big:
        tr.write_sep("!", "FULL-SIZE PARITY NOT CHECKED ON THIS HOST", red=True, bold=True)
        for r in big:
            tr.write_line(f"!!! SKIPPED (host memory): {r.nodeid}")
        tr.write_line(f"!!! {len(big)} full-size oracle comparisons did not run: the headline / C5 parity rows of DESIGN 5 are UNVERIFIED by this run "
                      "(the committed fixture tests/golden/c2_trajectory.npz still pins the headline: tests/test_gpu_c2_fixture.py needs no host memory).")
    if multi:
        tr.write_line(f"note: {len(multi)} multi-device tests skipped (one visible device): " + ", ".join(r.nodeid.split("::")[-1] for r in multi))
