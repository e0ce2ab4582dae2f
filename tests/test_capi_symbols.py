"""The C-ABI library loads without a GPU and exports every symbol include/transception_hip.h declares; the ctypes
binding (transception_amd/_lib.py) agrees with the header on every argument count.  No compute calls here."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "transception_hip.h")


def header_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    struct = re.search(r"typedef struct TcGemm \{(.*?)\} TcGemm;", src, flags=re.S).group(1)
    src = src.replace(struct, "")
    fns = {}
    for m in re.finditer(r"\b(?:int|long long)\s+(tc_\w+)\s*\(([^;]*?)\)\s*;", src, flags=re.S):
        args = m.group(2).strip()
        fns[m.group(1)] = 0 if args in ("", "void") else len([a for a in args.split(",") if a.strip()])
    nfields = sum(len([x for x in decl.split(",") if x.strip()]) for decl in
                  re.findall(r"(?:const void\*|void\*|float\*|int|long long|float)\s+([^;]+);", struct))
    return fns, nfields


@pytest.fixture(scope="module")
def built():
    from transception_amd.build import build
    return build(verbose=False)


def test_library_exports_every_declared_symbol(built):
    fns, _ = header_functions()
    assert len(fns) >= 30
    dll = ctypes.CDLL(built)
    for name in fns:
        assert hasattr(dll, name), f"{name} declared in the header but not exported by {built}"
    from transception_amd import _lib
    assert dll.tc_abi_version() == _lib.ABI_VERSION == 15


def test_ctypes_binding_matches_header(built):
    from transception_amd import _lib
    fns, nfields = header_functions()
    assert set(_lib.SIGNATURES) == set(fns), set(_lib.SIGNATURES) ^ set(fns)
    for name, n in fns.items():
        assert len(_lib.SIGNATURES[name]) == n, f"{name}: header has {n} args, ctypes binding {len(_lib.SIGNATURES[name])}"
    assert len(_lib.TcGemm._fields_) == nfields
    assert _lib.lib().tc_bn_scratch_floats(1000, 64) == 64 * (1 + 2 * 16)


def test_missing_library_fails_loudly(tmp_path):
    from transception_amd import _lib
    with pytest.raises(_lib.TcError):
        _lib._Lib(str(tmp_path / "nope.so"))
