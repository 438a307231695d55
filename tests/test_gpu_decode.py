"""D1 executes: csrc/rva_decode.hip and RocDecodeStream against the rocDecode test double (tests/mock_rocdecode/, built by
__graft_entry__.build()).  The real librocdecode.so is in neither image of this project, so this is the only way the
session code (callbacks, display queue, crop, hold / release, flush, mid-stream reconfigure) runs at all; it is labelled
test infrastructure and pins nothing about VCN pixels -- D1 stays "partial", bench.py keeps saying "not measured".  The
library is probed once per process, hence a worker process with RVA_ROCDECODE_LIB set (tests/decode_worker.py)."""
import os
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parents[1]


def test_rocdecode_session_runs_against_the_test_double():
    mock = ROOT / "tests" / "mock_rocdecode" / "libmockrocdecode.so"
    assert mock.exists(), "run __graft_entry__.build() first (it compiles the mock)"
    env = dict(os.environ, RVA_ROCDECODE_LIB=str(mock))
    p = subprocess.run([sys.executable, str(ROOT / "tests" / "decode_worker.py")], env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, text=True, timeout=600)
    assert p.returncode == 0 and "DECODE-WORKER-OK" in p.stdout, p.stdout[-4000:]
