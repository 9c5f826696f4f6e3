"""CPU checks of the boundary: the library builds/loads and exports every symbol the header declares;
the ctypes prototype table covers exactly the header.  No compute calls (no GPU here)."""
import os
import re

from rna_gan_amd import _abi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    src = open(os.path.join(ROOT, "include", "rnagan_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(rg_[a-z0-9_]+)\s*\(", src)))


def test_header_and_prototypes_agree():
    assert header_symbols() == sorted(_abi.PROTOTYPES.keys())


def test_library_loads_and_exports_all():
    if not os.path.exists(_abi.LIB_PATH):
        from rna_gan_amd.build import build_library
        build_library()
    lib = _abi.load()
    assert lib.rg_version() == _abi.ABI_VERSION
    for name in header_symbols():
        assert hasattr(lib, name), name


def test_product_does_not_import_oracle():
    pkg = os.path.join(ROOT, "rna_gan_amd")
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            txt = open(os.path.join(pkg, fn)).read()
            assert "import oracle" not in txt and "from oracle" not in txt, fn
