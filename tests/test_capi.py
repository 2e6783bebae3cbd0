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
    import re
    blob = open(_native.LIB_PATH, 'rb').read()
    # the offload bundle holds exactly one device code object, for gfx950: single target, no dual paths.  (The host side of
    # rocprim's radix sort carries a table of architecture NAMES for its tuning dispatch; names are not code objects.)
    targets = set(re.findall(rb'amdgcn-amd-amdhsa--(gfx[0-9a-f]+)', blob))
    assert targets == {b'gfx950'}, targets
    assert b'nvptx' not in blob and b'sm_80' not in blob and b'sm_90' not in blob


def test_product_package_never_touches_the_oracle():
    pkg = os.path.join(ROOT, 'ken-burns-effect_amd')
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(('.py', '.hip', '.h')):
                src = open(os.path.join(dirpath, f)).read()
                assert 'kbe_oracle' not in src.replace('oracle/kbe_oracle.c', '').replace('oracle/kbe_oracle', ''), f
                assert 'import oracle' not in src and 'from oracle' not in src, f
