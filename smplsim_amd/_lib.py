"""Loader of the HIP extension libsmplsim_hip.so (built in-tree by __graft_entry__.build()).

There is no CPU fallback: if the library is missing or cannot be loaded, importing the
stepper raises.  (The CPU oracle under oracle/ and the wavefront emulator under tests/ are
test infrastructure and are never reachable from here.)
"""
import ctypes
import os
import subprocess

from . import _cabi

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_PKG, "libsmplsim_hip.so")
SRC_DIR = os.path.join(_PKG, "csrc")
_LIB = None
# -fno-slp-vectorize: the SLP vectorizer turns the scalar FMA chains of the kernel into v_pk_* FP32 ops glued together
# with v_mov shuffles; packed FP32 issues at half rate on gfx950 (profiles/r01n_valu_ubench.txt), so that is slower
# (-7 %) and costs 45 more spilled VGPRs.  iterative-ilp: the GCN scheduler variant that schedules for ILP — the kernel
# is a dependent-instruction chain per wavefront with a fixed occupancy (launch bounds), not occupancy-limited (-3 %).
DEFAULT_OPT = "-O3 -fno-slp-vectorize -mllvm -amdgpu-sched-strategy=iterative-ilp"


class ExtensionMissing(RuntimeError):
    pass


def build(verbose=False, force=False):
    """hipcc --offload-arch=gfx950 build of the kernels + C ABI (cross-compiles without a GPU)."""
    srcs = [os.path.join(SRC_DIR, f) for f in ("smplsim_hip.hip", "ss_kernel.h", "ss_api.h", "ss_tables.h", "ss_hdr.h", "ss_motion.h",
                                                  "ss_motion_api.h")]
    srcs += [os.path.join(os.path.dirname(_PKG), "include", h) for h in ("smplsim_hip.h", "smplsim_motion.h")]
    opt = os.environ.get("SS_HIPCC_OPT", DEFAULT_OPT).split()
    stamp = LIB_PATH + ".flags"                              # rebuild when the flags change, not only the sources
    same_flags = os.path.exists(stamp) and open(stamp).read() == " ".join(opt)
    if not force and same_flags and os.path.exists(LIB_PATH) and all(os.path.getmtime(LIB_PATH) >= os.path.getmtime(s) for s in srcs):
        return LIB_PATH
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "--offload-arch=gfx950", *opt, "-std=c++17", "-shared", "-fPIC", srcs[0], "-o", LIB_PATH]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    with open(stamp, "w") as f:
        f.write(" ".join(opt))
    return LIB_PATH


def lib():
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise ExtensionMissing(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950). smplsim_amd has no CPU fallback.")
        _LIB = _cabi.bind(ctypes.CDLL(LIB_PATH))
    return _LIB
