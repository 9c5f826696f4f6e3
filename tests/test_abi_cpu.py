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
    assert lib.rg_storage_dtype() == _abi.RG_BF16
    for name in header_symbols():
        assert hasattr(lib, name), name


def test_fp16_build_loads_and_exports_all_but_the_probes():
    """librnagan_hip_f16.so: the same sources with IEEE fp16 as the 16-bit storage type (BASELINE configs[3]); it reports
    RG_F16 and exports every entry point of the header except the bf16-only measurement probes."""
    if not os.path.exists(_abi.LIB_PATH_F16):
        from rna_gan_amd.build import build_library
        build_library(half="f16")
    lib = _abi.load("f16")
    assert lib.rg_version() == _abi.ABI_VERSION
    assert lib.rg_storage_dtype() == _abi.RG_F16
    for name in header_symbols():
        if name.startswith(_abi.BF16_ONLY_PREFIXES):
            continue
        assert hasattr(lib, name), name


def test_product_does_not_import_oracle():
    pkg = os.path.join(ROOT, "rna_gan_amd")
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            txt = open(os.path.join(pkg, fn)).read()
            assert "import oracle" not in txt and "from oracle" not in txt, fn


def test_kernel_selection_options_are_known():
    """rg_set_option is host-only: every knob the header, the tools and the tests name is accepted (and cleared again),
    an unknown name is RG_EINVAL with a message."""
    lib = _abi.load()
    for name in (b"conv8", b"conv8_blocks", b"conv_tile", b"xcd", b"class_fast", b"wgrad_blocks", b"wgrad8", b"korder",
                 b"convp", b"convp_blocks", b"convd", b"convd_blocks", b"fp8_mx", b"wgrad8n", b"f32mma", b"skinny128", b"slab16", b"wslab16", b"bn_rev", b"wgrad8_mfma", b"narrow32", b"upimg", b"upimg_blocks"):
        assert lib.rg_set_option(name, 1) == 0, name
        assert lib.rg_set_option(name, -1) == 0, name
    assert lib.rg_set_option(b"no_such_knob", 1) != 0
    assert b"no_such_knob" in lib.rg_last_error()
