"""CPU: the C-ABI shared library loads and exports every symbol include/probpose_mi355x.h declares
(no compute calls -- there is no GPU here)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "probpose_mi355x.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pp_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(lib_built):
    lib = ctypes.CDLL(lib_built)
    names = _declared_symbols()
    assert "pp_probmap_decode" in names and "pp_last_error" in names
    for n in names:
        assert hasattr(lib, n), f"{n} declared in the header but not exported"


def test_python_binding_covers_header(lib_built):
    from probpose_code_amd import _lib

    assert sorted(_lib.SIGNATURES) == _declared_symbols()
    assert _lib.lib.pp_abi_version() == 1
    assert _lib.lib.pp_status_string(-2) == b"PP_ERR_UNSUPPORTED"


def test_argument_validation_without_gpu(lib_built):
    # argument checks run before any HIP call, so they are testable on the CPU box
    from probpose_code_amd import _lib

    st = _lib.lib.pp_probmap_decode(None, None, None, None, None, 1, 17, 64, 48, 192.0, 256.0,
                                    None, None, None, None, None, None)
    assert st == _lib.PP_ERR_INVALID_ARG
    assert b"non-NULL" in _lib.lib.pp_last_error()
