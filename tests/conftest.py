import json
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))
GOLDEN = ROOT / "tests" / "golden"


# Kernel selections persisted by earlier runs must not decide what a test validates: every test session tunes into its own
# directory (child processes inherit it).
import os
import tempfile
os.environ.setdefault("RVA_TUNE_CACHE_DIR", tempfile.mkdtemp(prefix="rva_tune_"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    return json.loads((GOLDEN / name).read_text())


@pytest.fixture(scope="session")
def post_cases():
    return load_golden("post_cases.json")["cases"]


@pytest.fixture(scope="session")
def tracker_cases():
    return load_golden("tracker_cases.json")
