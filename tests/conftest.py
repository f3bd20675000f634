"""Shared pytest configuration: markers, paths, golden-fixture loader."""

from __future__ import annotations

import sys
from pathlib import Path

import numpy as np
import pytest

REPO = Path(__file__).resolve().parent.parent
if str(REPO) not in sys.path:
    sys.path.insert(0, str(REPO))

GOLDEN = Path(__file__).resolve().parent / "golden"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name: str) -> dict:
    with np.load(GOLDEN / f"{name}.npz", allow_pickle=False) as z:
        return {k: z[k] for k in z.files}


@pytest.fixture(scope="session")
def golden():
    return load_golden


def canon_sign(c):
    """Largest-|component|-positive gauge for MO coefficient columns."""
    c = np.array(c, copy=True)
    idx = np.argmax(np.abs(c), axis=-2)
    sign = np.sign(np.take_along_axis(c, idx[..., None, :], axis=-2))
    sign[sign == 0] = 1
    return c * sign
