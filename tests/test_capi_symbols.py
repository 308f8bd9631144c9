"""The C-ABI shared library loads and exports every symbol include/blitzdg_hip.h declares
(no compute calls: runs without a GPU)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "blitzdg_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(bdg_[a-z0-9_]+)\s*\(", src)))


def test_header_declares_the_expected_groups():
    names = declared_symbols()
    assert "bdg_sw2d_rhs" in names and "bdg_mesh_read" in names and "bdg_trinodes_table" in names
    assert len(names) >= 45


def test_library_exports_every_declared_symbol():
    from blitzdg_amd import _capi
    lib = ctypes.CDLL(_capi.LIB_PATH)
    missing = [n for n in declared_symbols() if not hasattr(lib, n)]
    assert not missing, f"not exported: {missing}"


def test_python_binding_covers_every_declared_symbol():
    from blitzdg_amd import _capi
    assert sorted(_capi.EXPORTED_SYMBOLS) == declared_symbols()


def test_version_and_error_string():
    from blitzdg_amd import _capi
    assert _capi.lib.bdg_version() >= 100
    with pytest.raises(_capi.BdgError) as e:
        _capi.check(_capi.lib.bdg_mesh_read(None, b"nope"))
    assert e.value.code == _capi.BDG_ERR_ARGUMENT


def test_sw2d_create_rejects_bad_arguments_without_touching_a_gpu():
    from blitzdg_amd import _capi
    h = ctypes.c_void_p()
    d = _capi.Sw2dDesc()
    d.order = 99
    rc = _capi.lib.bdg_sw2d_create(ctypes.byref(d), ctypes.byref(h))
    assert rc == _capi.BDG_ERR_ARGUMENT and not h.value
    assert _capi.lib.bdg_sw2d_create(None, ctypes.byref(h)) == _capi.BDG_ERR_ARGUMENT
    # NULL handles are reported, not dereferenced
    assert _capi.lib.bdg_sw2d_step_lserk4(None, 0.1, 1) == _capi.BDG_ERR_ARGUMENT
    assert _capi.lib.bdg_sw2d_device_bytes(None) == 0


def test_oracle_is_not_referenced_by_the_product():
    """The product package must never import or link the CPU oracle."""
    pkg = os.path.join(ROOT, "blitzdg_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hpp", ".hip", ".h")) or f == "Makefile":
                text = open(os.path.join(dirpath, f), errors="replace").read()
                assert "oracle" not in text.lower() or f == "__init__.py" and "oracle" not in text.lower(), \
                    f"{os.path.join(dirpath, f)} mentions the oracle"
