"""CPU: the C-ABI library builds in-tree, loads, reports the ABI version, and exports every symbol
include/cclip_hip.h declares (no compute calls: there is no GPU here)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    hdr = open(os.path.join(ROOT, "include", "cclip_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\bint\s+(cclip_\w+)\s*\(", hdr)))


def test_library_exports_every_declared_symbol():
    import __graft_entry__ as ge
    ge.build()
    from cclip_hip import LIB_PATH, load_library
    assert os.path.exists(LIB_PATH) and LIB_PATH.startswith(ROOT), "the .so must live in-tree"
    lib = load_library()
    names = _declared()
    assert len(names) >= 19, names
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/cclip_hip.h but not exported"
    assert lib.cclip_abi_version() == 3


def test_no_torch_types_in_abi():
    hdr = open(os.path.join(ROOT, "include", "cclip_hip.h")).read()
    assert "torch" not in re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    assert "at::" not in hdr and "c10::" not in hdr


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    import cclip_hip._lib as L
    monkeypatch.setattr(L, "_lib", None)
    monkeypatch.setattr(L, "LIB_PATH", str(tmp_path / "nope.so"))
    import pytest
    with pytest.raises(RuntimeError, match="no non-HIP compute path"):
        L.load_library()


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "construction-clip_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(dp, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f"{f} imports the oracle"
