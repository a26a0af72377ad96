import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

os.environ.setdefault("OMP_NUM_THREADS", "2")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run through gpurun)")


@pytest.fixture
def tmp_dirs(tmp_path, monkeypatch):
    monkeypatch.setenv("DDL_LOG_DIR", str(tmp_path / "logs"))
    monkeypatch.setenv("DDL_CKPT_DIR", str(tmp_path / "ckpt"))
    monkeypatch.setenv("DDL_DATA_DIR", str(tmp_path / "nodata"))
    return tmp_path
