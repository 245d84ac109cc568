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
# SMPLSIM_HIP_LIB: another build of the same library (A/B of compiler flags or kernel variants, tools/build_variant.sh)
LIB_PATH = os.environ.get("SMPLSIM_HIP_LIB") or os.path.join(_PKG, "libsmplsim_hip.so")
SRC_DIR = os.path.join(_PKG, "csrc")
_LIB = None
# -fno-slp-vectorize: the SLP vectorizer turns the scalar FMA chains of the kernel into v_pk_* FP32 ops glued together
# with v_mov shuffles; packed FP32 issues at half rate on gfx950 (profiles/r01n_valu_ubench.txt), so that is slower
# (-7 %) and costs 45 more spilled VGPRs.  iterative-ilp: the GCN scheduler variant that schedules for ILP — the kernel
# is a dependent-instruction chain per wavefront with a fixed occupancy (launch bounds), not occupancy-limited (-3 %).
# -Os (was -O3): the stepper is one big state machine per wavefront; optimised for size it spills more (47 instead of 16
# VGPRs) and is still faster — 2.04 vs 2.08 ms per 4096-env SMPL step, 4.80 vs 5.22 ms SMPL-X (same-box A/B; -O2 is in
# between, -Oz and -O3 -fno-unroll-loops are slower).  The small arithmetic kernels of the motion library are twice as slow at
# -Os, so that translation unit keeps -O3 (MOTION_OPT).
DEFAULT_OPT = "-Os -fno-slp-vectorize -mllvm -amdgpu-sched-strategy=iterative-ilp"
# the body-body-contact instantiations (smplsim_hip_sc.hip; round 4: dense block solve over the coupled set) are the other way round:
# -O3 / -O2 5.90 ms per 4096-env step, -Os 6.39 ms (same-box A/B, profiles/r04_selfcol_ab.txt); scratch 912 vs 1104 bytes per lane.
# Round 6 (dense solve on the matrix core): -O2 4.32 ms, -O3 4.40, -Os 4.41, -O3 with the default scheduler 4.62 (profiles/r06_selfcol_ab.txt)
SC_OPT = "-O2 -fno-slp-vectorize -mllvm -amdgpu-sched-strategy=iterative-ilp"
MOTION_OPT = "-O3"
# the SMPL-X/H size class (smplsim_hip_x.hip, round 5; 256 VGPRs; A/B taken at 6 envs per CU, 7 since the lean tables): -O2 2.93 ms per 4096-env step, -O3 2.95, -Os 2.97
X_OPT = "-O2 -fno-slp-vectorize -mllvm -amdgpu-sched-strategy=iterative-ilp"


class ExtensionMissing(RuntimeError):
    pass


def build(verbose=False, force=False):
    """hipcc --offload-arch=gfx950 build of the kernels + C ABI (cross-compiles without a GPU)."""
    srcs = [os.path.join(SRC_DIR, f) for f in ("smplsim_hip.hip", "smplsim_hip_sc.hip", "smplsim_hip_im.hip", "smplsim_motion.hip", "smplsim_mlp.hip", "smplsim_hip_x.hip", "ss_env_kernel.h", "ss_kernel.h", "ss_selfcol.h", "ss_api.h", "ss_tables.h", "ss_hdr.h",
                                                  "ss_motion.h", "ss_motion_api.h", "ss_wave_gpu.h", "ss_imfused.h", "ss_mjcf.h", "ss_gemm256.h")]
    srcs += [os.path.join(os.path.dirname(_PKG), "include", h) for h in ("smplsim_hip.h", "smplsim_motion.h", "smplsim_mlp.h")]
    opt = os.environ.get("SS_HIPCC_OPT", DEFAULT_OPT).split()
    scopt = os.environ.get("SS_HIPCC_SC_OPT", SC_OPT).split()
    mopt = os.environ.get("SS_HIPCC_MOTION_OPT", MOTION_OPT).split()
    xopt = os.environ.get("SS_HIPCC_X_OPT", X_OPT).split()
    stamp = LIB_PATH + ".flags"                              # rebuild when the flags change, not only the sources
    flag_str = " ".join(opt + ["|"] + scopt + ["|"] + mopt + ["|"] + xopt)
    same_flags = os.path.exists(stamp) and open(stamp).read() == flag_str
    if not force and same_flags and os.path.exists(LIB_PATH) and all(os.path.getmtime(LIB_PATH) >= os.path.getmtime(s) for s in srcs):
        return LIB_PATH
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs, procs = [], []
    for src, flags in ((srcs[0], opt), (srcs[1], scopt), (srcs[2], opt), (srcs[3], mopt), (srcs[4], mopt), (srcs[5], xopt)):   # stepper (4 units), motion library, policy MLP: their own
        obj = os.path.join(_PKG, os.path.basename(src).replace(".hip", ".o"))   # flags, compiled side by side
        cmd = [hipcc, "--offload-arch=gfx950", *flags, "-std=c++17", "-fPIC", "-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd))
        procs.append((cmd, subprocess.Popen(cmd)))
        objs.append(obj)
    for cmd, pr in procs:
        if pr.wait() != 0:
            raise subprocess.CalledProcessError(pr.returncode, cmd)
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", LIB_PATH]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    for o in objs:
        os.remove(o)
    with open(stamp, "w") as f:
        f.write(flag_str)
    return LIB_PATH


def source_hash():
    """What a profile / counter summary was taken on: sha256 (16 hex digits) over every file under smplsim_amd/csrc and include/
    plus the compiler flags of the three groups of translation units.  tools/prof_summarize.py stores it in profiles/pmc_summary_*.json,
    bench.py recomputes it and refuses counter blocks of another tree (roofline.pmc_stale)."""
    import hashlib
    h = hashlib.sha256()
    inc = os.path.join(os.path.dirname(_PKG), "include")
    for d in (SRC_DIR, inc):
        for f in sorted(os.listdir(d)):
            if f.endswith((".h", ".hip")):
                h.update(f.encode()); h.update(open(os.path.join(d, f), "rb").read())
    h.update(" ".join([os.environ.get("SS_HIPCC_OPT", DEFAULT_OPT), "|", os.environ.get("SS_HIPCC_SC_OPT", SC_OPT), "|",
                       os.environ.get("SS_HIPCC_MOTION_OPT", MOTION_OPT), "|", os.environ.get("SS_HIPCC_X_OPT", X_OPT)]).encode())
    return h.hexdigest()[:16]


def lib():
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise ExtensionMissing(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950). smplsim_amd has no CPU fallback.")
        # torch first: it brings its own libamdhip64; loaded after ours (which links /opt/rocm's) the process would hold two HIP
        # runtimes and the first hipSetDevice fails ("cannot select device")
        import torch  # noqa: F401
        _LIB = _cabi.bind_mlp(_cabi.bind(ctypes.CDLL(LIB_PATH)))
    return _LIB
