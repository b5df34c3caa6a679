"""CPU-side checks of the C-ABI: the library loads and exports every symbol include/tsc.h
declares (no compute calls -- there is no GPU here)."""
import ctypes
import os
import re

import pytest

from deeprl_signal_control_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, 'include', 'tsc.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(tsc_[a-z0-9_]+)\s*\(', src)))


def test_header_symbols_exported():
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    L = ctypes.CDLL(_lib.LIB_PATH)
    names = _declared()
    assert len(names) >= 10
    for n in names:
        assert hasattr(L, n), 'libtsc.so does not export %s' % n
    assert sorted(_lib.SYMBOLS) == names
    assert L.tsc_version() >= 100


def test_product_never_imports_oracle():
    """The product path must not reach into oracle/ (tier rule 3)."""
    pkg = os.path.join(ROOT, 'deeprl_signal_control_amd')
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(('.py', '.hip', '.h', '.cpp')):
                txt = open(os.path.join(dp, f)).read()
                assert not re.search(r'^\s*(from|import)\s+oracle', txt, re.M), f
                assert '#include "../../oracle' not in txt and 'oracle/' not in txt.replace('CPU oracle', ''), f


def test_missing_library_fails_loudly(monkeypatch):
    monkeypatch.setattr(_lib, '_LIB', None)
    monkeypatch.setattr(_lib, 'LIB_PATH', '/nonexistent/libtsc.so')
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        _lib.lib()
