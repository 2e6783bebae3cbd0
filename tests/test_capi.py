"""The C-ABI library loads without a GPU and exports every symbol include/kbe.h declares."""
import ctypes
import os
import re

from conftest import ROOT


def _declared():
    text = open(os.path.join(ROOT, 'include', 'kbe.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(kbe_[a-z0-9_]+)\s*\(', text)))


def test_header_and_binding_agree():
    from ken_burns_effect_amd import _native
    assert _declared() == sorted(_native.SYMBOLS)


def test_library_exports_every_declared_symbol():
    from ken_burns_effect_amd import _native
    assert os.path.exists(_native.LIB_PATH), 'run __graft_entry__.build() first'
    lib = ctypes.CDLL(_native.LIB_PATH)
    for name in _declared():
        assert hasattr(lib, name), name
    lib.kbe_abi_version.restype = ctypes.c_int
    assert lib.kbe_abi_version() == _native.ABI_VERSION
    assert _native.load() is not None


def test_library_is_a_gfx950_code_object():
    from ken_burns_effect_amd import _native
    blob = open(_native.LIB_PATH, 'rb').read()
    assert b'gfx950' in blob
    assert b'gfx942' not in blob and b'sm_' not in blob      # single target, no dual paths


def test_product_package_never_touches_the_oracle():
    pkg = os.path.join(ROOT, 'ken-burns-effect_amd')
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(('.py', '.hip', '.h')):
                src = open(os.path.join(dirpath, f)).read()
                assert 'kbe_oracle' not in src.replace('oracle/kbe_oracle.c', '').replace('oracle/kbe_oracle', ''), f
                assert 'import oracle' not in src and 'from oracle' not in src, f
