"""CPU checks of the C-ABI boundary: the library loads and exports every symbol
include/algp_hip.h declares; no compute call is made (no GPU here)."""
import os
import re

import pytest

from algp_amd import _hip

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    txt = open(os.path.join(REPO, 'include', 'algp_hip.h')).read()
    txt = re.sub(r'/\*.*?\*/', '', txt, flags=re.S)
    return sorted(set(re.findall(r'\b(algp_[a-z0-9_]+)\s*\(', txt)))


def test_header_and_binding_agree():
    decl = _declared_symbols()
    assert len(decl) >= 35
    assert sorted(_hip.SIGNATURES) == decl


def test_library_exports_every_declared_symbol():
    lib = _hip.load()
    for name in _declared_symbols():
        assert hasattr(lib, name), name
    assert lib.algp_version() >= 100


def test_no_silent_fallback_without_gpu():
    if _hip.device_count() > 0:
        pytest.skip('a GPU is visible')
    with pytest.raises(_hip.AlgpError):
        _hip.Context()


def test_product_never_imports_oracle():
    pkg = os.path.join(REPO, 'algp_amd')
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(('.py', '.hip', '.h', '.cpp')):
                src = open(os.path.join(root, f)).read()
                assert 'import oracle' not in src and 'from oracle' not in src, f
