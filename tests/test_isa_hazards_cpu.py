"""Static check of the generated gfx950 code for hazards around INLINE-ASM instructions, which hipcc's hazard recognizer does not
see (tools/isa_hazards.py).  rg_convp.hip issues its 16-byte stores as inline asm: on gfx940 and later a VALU write into such a
store's data registers needs two wait states behind the store.  Round 6 found the fp16 build's statistics variant violating it
(garbage in the first channel pair of every pixel); every asm store now carries its own wait states and this test keeps it so in
both builds of the library."""
import os
import shutil
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import isa_hazards  # noqa: E402

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


@pytest.mark.skipif(not (os.path.exists(HIPCC) or shutil.which("hipcc")), reason="hipcc not available")
@pytest.mark.parametrize("defines", [(), ("-DRG_HALF_F16=1",)], ids=["bf16", "f16"])
def test_convp_asm_stores_keep_their_wait_states(tmp_path, defines):
    src = os.path.join(ROOT, "rna_gan_amd", "csrc", "rg_convp.hip")
    out = str(tmp_path / "convp.s")
    isa_hazards.compile_to_asm(src, out, defines)
    n_stores = sum(1 for _, _, op, _, _ in isa_hazards.instructions(out) if op.startswith(isa_hazards.WIDE_STORES))
    assert n_stores >= 12 * 16, "the kernel's asm stores were not found in the listing (%d)" % n_stores
    flags = isa_hazards.store_hazards(out, wait=2)
    assert not flags, "VALU write into the data registers of a 16-byte store within 2 wait states: %r" % (flags[:3],)


def test_store_hazard_checker_flags_a_known_bad_sequence(tmp_path):
    p = tmp_path / "bad.s"
    p.write_text("_Zk:\n\tglobal_store_dwordx4 v1, v[4:7], s[0:1]\n\tv_or_b32_e32 v4, v9, v8\n"
                 "\tglobal_store_dwordx4 v1, v[4:7], s[0:1]\n\ts_nop 1\n\tv_or_b32_e32 v4, v9, v8\n"
                 "\tglobal_store_dwordx2 v1, v[4:5], s[0:1]\n\tv_mov_b32_e32 v4, 0\n")
    flags = isa_hazards.store_hazards(str(p), wait=2)
    assert len(flags) == 1 and flags[0][1] == 3 and flags[0][4] == 0
