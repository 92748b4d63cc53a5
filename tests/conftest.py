"""Shared test plumbing: loads the product binding (game-engine_amd/__init__.py -> libgame_hip.so)
and the CPU oracle (oracle/libgame_oracle.so, test infrastructure only)."""
import importlib.util
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_product():
    if "game_engine_amd" in sys.modules:
        return sys.modules["game_engine_amd"]
    path = os.path.join(ROOT, "game-engine_amd", "__init__.py")
    spec = importlib.util.spec_from_file_location("game_engine_amd", path,
                                                  submodule_search_locations=[os.path.dirname(path)])
    mod = importlib.util.module_from_spec(spec)
    sys.modules["game_engine_amd"] = mod
    spec.loader.exec_module(mod)
    return mod


def _load_oracle_binding():
    """oracle/binding.py: ctypes view of the CPU restatement (test infrastructure)."""
    if "game_oracle_binding" not in sys.modules:
        spec = importlib.util.spec_from_file_location("game_oracle_binding", os.path.join(ROOT, "oracle", "binding.py"))
        mod = importlib.util.module_from_spec(spec)
        sys.modules["game_oracle_binding"] = mod
        spec.loader.exec_module(mod)
    return sys.modules["game_oracle_binding"]


Oracle = _load_oracle_binding().Oracle


@pytest.fixture(scope="session")
def game():
    return load_product()


@pytest.fixture(scope="session")
def oracle():
    return Oracle()


REFERENCE = "/root/reference"


def reference_available():
    return os.path.isdir(REFERENCE)


# The benchmark cinematics (the GPU box has no copy of the reference tree) live with the product: game-engine_amd/workloads.py.
_W = load_product().workloads
C1_HELLO, C2_FBM, C2B_NOISE, C3_SHOWCASE, C4_TRAILS, C5_GRAVITY = (_W.C1_HELLO, _W.C2_FBM, _W.C2B_NOISE, _W.C3_SHOWCASE,
                                                                   _W.C4_TRAILS, _W.C5_GRAVITY)
