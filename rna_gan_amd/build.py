"""Build librnagan_hip.so (hipcc, gfx950 only) in-tree: rna_gan_amd/librnagan_hip.so.

hipcc cross-compiles without a GPU, so this runs in the build container; the resulting .so
travels to the GPU box with the repo snapshot (it is git-ignored, not gpurun-ignored).
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "librnagan_hip.so")
OUT_F16 = os.path.join(HERE, "librnagan_hip_f16.so")
OBJ = os.path.join(HERE, "csrc", "_obj")
SOURCES = ["rg_api.hip", "rg_generic.hip", "rg_bn.hip", "rg_misc.hip", "rg_mfma.hip", "rg_conv8.hip", "rg_convp.hip", "rg_convd.hip", "rg_wgrad8.hip", "rg_skinny.hip", "rg_vae.hip", "rg_splitbn.hip", "rg_incep.hip", "rg_g0adam.hip", "rg_upimg.hip", "rg_probe.hip", "rg_conv8f.hip", "rg_wgrad8f.hip"]
# sources a translation unit #includes besides the headers (rg_probe.hip instantiates the product's conv8_kernel template with
# its measurement flag from the same source text)
EXTRA_DEPS = {"rg_probe.hip": ["rg_conv8.hip"], "rg_conv8f.hip": ["rg_conv8.hip"], "rg_wgrad8f.hip": ["rg_wgrad8.hip"]}
BF16_ONLY = ("rg_probe.hip", "rg_conv8f.hip", "rg_wgrad8f.hip")      # written for bf16 operands (measurement probes; the fp32 mode's bf16-plane kernels): not part of the fp16 build
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-function",
         "-ffp-contract=off"]


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build_library(force=False, verbose=True, half="bf16"):
    """half = "bf16": rna_gan_amd/librnagan_hip.so.  half = "f16": the SAME sources with -DRG_HALF_F16 (the library's 16-bit
    storage type is IEEE fp16, rg_common.h) -> rna_gan_amd/librnagan_hip_f16.so, without the measurement probes."""
    if half not in ("bf16", "f16"):
        raise ValueError("half must be 'bf16' or 'f16'")
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not os.path.exists(hipcc):
        hipcc = "hipcc"
    f16 = half == "f16"
    objdir = OBJ + ("_f16" if f16 else "")
    out = OUT_F16 if f16 else OUT
    sources = [s for s in SOURCES if not (f16 and s in BF16_ONLY)]
    flags = FLAGS + (["-DRG_HALF_F16=1"] if f16 else [])
    os.makedirs(objdir, exist_ok=True)
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    headers.append(os.path.join(os.path.dirname(HERE), "include", "rnagan_hip.h"))
    jobs = []
    for s in sources:
        src = os.path.join(CSRC, s)
        obj = os.path.join(objdir, s.replace(".hip", ".o"))
        if force or _stale(obj, [src] + headers + [os.path.join(CSRC, d) for d in EXTRA_DEPS.get(s, [])]):
            jobs.append((src, obj))

    def cc(job):
        src, obj = job
        cmd = [hipcc] + flags + ["-c", src, "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed for %s:\n%s" % (src, r.stderr[-4000:]))
        if verbose and r.stderr.strip():
            sys.stderr.write(r.stderr[-2000:])
        return obj

    with ThreadPoolExecutor(max_workers=6) as ex:
        list(ex.map(cc, jobs))
    objs = [os.path.join(objdir, s.replace(".hip", ".o")) for s in sources]
    if force or jobs or _stale(out, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n%s" % r.stderr[-4000:])
    return out


def build_all(force=False, verbose=True):
    """Both builds of the library (what __graft_entry__.build() runs)."""
    return [build_library(force, verbose, "bf16"), build_library(force, verbose, "f16")]


def build_debug_library(force=False):
    """tools/debug/librnagan_debug.so: diagnostic kernels that are NOT part of the product ABI (the CU-holding stand-in for a
    collective, RNAGAN_DEBUG_HOG).  Built on demand; never loaded by a product path."""
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not os.path.exists(hipcc):
        hipcc = "hipcc"
    d = os.path.join(os.path.dirname(HERE), "tools", "debug")
    src = os.path.join(d, "rg_debug_hold.hip")
    # several ranks may get here at once (RNAGAN_DEBUG_HOG under torchrun): each compiles to a file of its own and renames
    # it into place -- a rank never dlopens a half-written library; a read-only tree uses a per-user cache (the output path is
    # chosen FIRST, so that the staleness check looks at the copy that will be loaded and a cached build is reused)
    d_out = d if os.access(d, os.W_OK) else os.path.join(os.path.expanduser("~"), ".cache", "rna_gan_amd")
    os.makedirs(d_out, exist_ok=True)
    out = os.path.join(d_out, "librnagan_debug.so")
    if force or _stale(out, [src]):
        tmp = "%s.%d.tmp" % (out, os.getpid())
        r = subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-fPIC", "-shared", "-o", tmp, src], capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed for %s:\n%s" % (src, r.stderr[-4000:]))
        os.replace(tmp, out)
    return out


if __name__ == "__main__":
    print("\n".join(build_all(force="--force" in sys.argv)))
