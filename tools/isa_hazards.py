#!/usr/bin/env python3
"""Static checks of gfx950 assembly for hazards hipcc cannot see around INLINE-ASM instructions (rg_convp.hip issues its MFMAs,
LDS reads and 16-byte stores as inline asm; the compiler's hazard recognizer treats an asm block as opaque).

  store : a VALU write into the data registers of a global / buffer store of more than 8 bytes within WAIT wait states behind the
          store (gfx940 and later need 2; found in round 6: the fp16 build's statistics variant of convp_kernel rebuilt o[0] in
          the instruction behind its store and wrote garbage -- the bf16 build happened to schedule another instruction between)
  mfma  : a non-MFMA write into an A / B operand register of an MFMA that started fewer than BUSY cycles ago (crude issue model:
          one cycle per instruction, s_nop N = N + 1; borderline flags at BUSY - 2 .. BUSY are expected)

    python tools/isa_hazards.py store FILE.s [WAIT=3]
    python tools/isa_hazards.py mfma FILE.s [KERNEL-SUBSTRING] [BUSY=16]
    python tools/isa_hazards.py compile rna_gan_amd/csrc/rg_convp.hip OUT.s [-DRG_HALF_F16=1]     # hipcc -S, device only
"""
import os
import re
import subprocess
import sys

_REG = re.compile(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b")
WIDE_STORES = ("global_store_dwordx3", "global_store_dwordx4", "buffer_store_dwordx3", "buffer_store_dwordx4",
               "flat_store_dwordx3", "flat_store_dwordx4")
STORES = ("global_store", "buffer_store", "ds_write", "ds_store", "flat_store", "scratch_store")


def vregs(tok):
    out = set()
    for m in _REG.finditer(tok):
        if m.group(1):
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
        else:
            out.add(int(m.group(3)))
    return out


def instructions(path):
    """(line number, kernel label, mnemonic, operand list) of every instruction."""
    cur = None
    for ln, line in enumerate(open(path), 1):
        s = line.split(";")[0].strip()
        if not s or s.startswith("."):
            continue
        if s.endswith(":"):
            if s.startswith("_Z"):
                cur = s[:-1]
            continue
        op, _, rest = s.partition(" ")
        yield ln, cur, op, [o.strip() for o in rest.split(",")], s


def store_hazards(path, wait=3):
    """[(kernel, line, text, store line, wait states given)]"""
    flags, pend, last = [], [], None
    for ln, cur, op, ops, s in instructions(path):
        if cur != last:
            pend, last = [], cur
        cost = int(ops[0], 0) + 1 if op == "s_nop" else 1
        if op.startswith(WIDE_STORES):
            pend = [(0, vregs(ops[1]), ln)] + [(a + 1, d, l) for a, d, l in pend if a + 1 < wait]
            continue
        writes = set() if (op.startswith(STORES) or op.startswith("s_")) else vregs(ops[0])
        for a, d, l in pend:
            if writes & d:
                flags.append((cur, ln, s, l, a))
        pend = [(a + cost, d, l) for a, d, l in pend if a + cost < wait]
    return flags


def mfma_hazards(path, want="", busy=16):
    flags, live, t, last_start, last = [], [], 0, -100, None
    for ln, cur, op, ops, s in instructions(path):
        if cur != last:
            live, t, last_start, last = [], 0, -100, cur
        if cur is None or want not in cur:
            continue
        if op.startswith("v_mfma"):
            start = max(t, last_start + busy)
            last_start, t = start, start + 1
            live = (live + [(start, vregs(ops[1]) | vregs(ops[2]), ln)])[-40:]
            continue
        if op == "s_nop":
            t += int(ops[0], 0) + 1
            continue
        t += 1
        if op.startswith("s_") or op.startswith(STORES):
            continue
        writes = vregs(ops[0])
        for st, ab, l0 in live:
            if t < st + busy and writes & ab:
                flags.append((cur, ln, s, l0, t - st))
    return flags


def compile_to_asm(src, out, defines=()):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-ffp-contract=off", "-S", "--cuda-device-only"]
    r = subprocess.run(cmd + list(defines) + [src, "-o", out], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc -S failed:\n" + r.stderr[-3000:])
    return out


if __name__ == "__main__":
    if len(sys.argv) < 3:
        sys.exit(__doc__)
    what = sys.argv[1]
    if what == "compile":
        print(compile_to_asm(sys.argv[2], sys.argv[3], sys.argv[4:]))
    elif what == "store":
        fl = store_hazards(sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 3)
        for k, ln, s, l, a in fl:
            print("%s line %d: %s  overwrites data of the store at line %d (%d wait states behind it)" % (k[:60], ln, s, l, a))
        print("store hazards:", len(fl))
    elif what == "mfma":
        fl = mfma_hazards(sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else "", int(sys.argv[4]) if len(sys.argv) > 4 else 16)
        for k, ln, s, l, a in fl:
            print("%s line %d: %s  writes an operand of the MFMA at line %d (%d cycles after its start)" % (k[:60], ln, s, l, a))
        print("mfma operand hazards:", len(fl))
    else:
        sys.exit(__doc__)
