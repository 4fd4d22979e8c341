"""Shared pytest setup.  Markers: ``gpu`` = needs a real MI355X (run with -m gpu)."""
import sys
from pathlib import Path

import pytest

REPO = Path(__file__).resolve().parents[1]
if str(REPO) not in sys.path:
    sys.path.insert(0, str(REPO))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real MI355X GPU (HIP path parity)")


@pytest.fixture(scope="session")
def golden():
    from safetensors import safe_open
    from tests.golden import inputs as gi

    class Golden:
        manifest = gi.load_manifest()

        def __init__(self):
            self._files = {}

        def get(self, fname, key):
            if fname not in self._files:
                store = {}
                with safe_open(str(gi.GOLDEN_DIR / fname), framework="pt") as f:
                    for k in f.keys():
                        store[k] = f.get_tensor(k)
                self._files[fname] = store
            return self._files[fname][key]

        def keys(self, fname):
            self.get(fname, next(iter(self._peek(fname))))
            return list(self._files[fname].keys())

        def _peek(self, fname):
            with safe_open(str(gi.GOLDEN_DIR / fname), framework="pt") as f:
                return list(f.keys())

    return Golden()
