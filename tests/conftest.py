"""pytest configuration: markers + import paths.

`-m "not gpu"`: oracle vs golden vectors, host logic, C-ABI symbol checks (CPU).
`-m gpu`     : parity tests proper -- HIP path through the C-ABI vs the oracle.
"""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
for p in (ROOT / "tests", ROOT / "seq-align_amd" / "python", ROOT):
    if str(p) not in sys.path:
        sys.path.insert(0, str(p))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run via gpurun)")
