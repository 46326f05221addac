"""Build recipe for the HIP engine: hipcc --offload-arch=gfx950, one object per translation unit
(in parallel), linked into pytorch_mppi_amd/libmppi_amd.so next to this file (in-tree, so the
library travels to the GPU box with the repo snapshot).  gfx950 only -- no other arch, no
compatibility paths."""
import concurrent.futures as cf
import hashlib
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
INCLUDE = os.path.join(os.path.dirname(HERE), "include")
# MPPI_LIB_SUFFIX: experiment builds (tools/) next to the product library
SUFFIX = os.environ.get("MPPI_LIB_SUFFIX", "")
LIB = os.path.join(HERE, f"libmppi_amd{SUFFIX}.so")
OBJ_DIR = os.path.join(CSRC, "build" + SUFFIX)
SOURCES = ["capi.hip", "dist.hip", "group.hip", "update.hip", "rollout_pendulum.hip", "rollout_integrator.hip",
           "rollout_linear_goal.hip", "rollout_mlp.hip", "rollout_mlp_mfma.hip", "rollout_mlp_split.hip",
                   "noise_torch.hip"]
# translation units: (source, object name, extra flags).  The two heaviest sources are compiled as several units each
# (groups of model dimensions selected with a define) so that the parallel build is not one long compile
_GROUPS = {"rollout_integrator.hip": ("MPPI_INTEGRATOR_GROUP", 4), "rollout_linear_goal.hip": ("MPPI_LINEAR_GROUP", 3),
           "rollout_mlp.hip": ("MPPI_MLP_GROUP", 6), "rollout_mlp_split.hip": ("MPPI_SPLIT_GROUP", 2)}


def _units():
    units = []
    for src in SOURCES:
        if src in _GROUPS:
            for g in range(_GROUPS[src][1]):
                units.append((src, src.replace(".hip", f"_g{g}.o"), [f"-D{_GROUPS[src][0]}={g}"]))
        else:
            units.append((src, src.replace(".hip", ".o"), []))
    # longest first (measured seconds per unit on the build container, 8 at a time): the pool starts the big ones first
    cost = {"update.o": 165, "rollout_linear_goal_g2.o": 126, "rollout_integrator_g2.o": 90,
            "rollout_integrator_g0.o": 88,
            "rollout_integrator_g1.o": 83, "rollout_linear_goal_g0.o": 79, "rollout_mlp_g0.o": 78,
                    "rollout_mlp_g1.o": 75,
            "rollout_mlp_g2.o": 75, "rollout_pendulum.o": 70, "rollout_linear_goal_g1.o": 68,
                    "rollout_integrator_g3.o": 45,
            "rollout_mlp_g3.o": 75, "rollout_mlp_g4.o": 75, "rollout_mlp_g5.o": 75, "rollout_mlp_split_g0.o": 30,
            "rollout_mlp_split_g1.o": 35}
    return sorted(units, key=lambda u: -cost.get(u[1], 5))
# -ffp-contract=fast: mul+add pairs fuse into v_fma / v_pk_fma.  torch eager rounds twice where
# the kernels round once, a <= 1 ulp difference per operation that the parity tests bound
# (1e-5 relative fp32, 1e-9 fp64 on every public output of command()).
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=fast",
         "-Wno-unused-result"] + os.environ.get("MPPI_EXTRA_HIPCC_FLAGS", "").split()


# per-source extras: the Philox instantiations of K3 only stay scratch-free when their 16-row
# tile loop is fully unrolled (static accumulator indices), which needs a higher pragma-unroll budget
EXTRA = {"update.hip": ["-mllvm", "-pragma-unroll-threshold=200000"],
         # MFMA results straight into VGPRs: every hidden activation is read by the VALU (tanh), and an
         # AGPR accumulator costs a v_accvgpr_read per value in a VALU-bound kernel
         # (-fno-slp-vectorize: the fp16 residual of the activation split is one v_fma_mix_f32 per value only when the
         # SLP vectorizer does not pair two of them into v_cvt_f32_f16 x 2 + v_pk_fma_f32 first)
         "rollout_mlp_split.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form", "-fno-slp-vectorize"],
         # the KMPPI-fused K1 (rollout_kmppi.hpp) pins 256 control points in the AGPRs and reads its 4 x nu
         # accumulator tile with the VALU: same flag for every unit that instantiates it (jit.py passes it too)
         "rollout_integrator.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"],
         "rollout_linear_goal.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"],
         # the torch.randn stream, bit for bit: rocrand's Box-Muller with the contraction rule the library itself is
         # built with (a later -ffp-contract wins; with =fast one value in ~10^5 differs from torch's in its last bit)
         "noise_torch.hip": ["-ffp-contract=on"]}
K1_FLAGS = ["-mllvm", "-amdgpu-mfma-vgpr-form"]     # for translation units built around csrc/rollout.hpp (jit.py)
# heavy user models (jit.py) only: the SLP vectorizer pairs the products of a traced network into v_pk_mul_f32 before fp
# contraction sees them (570 mul + add pairs instead of fmas in a 3700-operation step); packed fp32 arithmetic is no
# faster than two scalar instructions on gfx950.  Light models keep the flags of the built-in units (a snippet model
# that restates a built-in one compiles to the same kernel, bit for bit: tests/test_gpu_jit_models.py)
K1_HEAVY_FLAGS = ["-fno-slp-vectorize"]


def _hipcc():
    for c in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found: the MPPI engine has no non-HIP build")


def _deps_hash():
    h = hashlib.sha256()
    names = sorted(os.listdir(CSRC))
    for n in names:
        if n.endswith((".hip", ".hpp", ".h")):
            h.update(n.encode())
            h.update(open(os.path.join(CSRC, n), "rb").read())
    h.update(open(os.path.join(INCLUDE, "mppi_amd.h"), "rb").read())
    h.update(" ".join(FLAGS).encode())
    h.update(repr(sorted(EXTRA.items())).encode())
    h.update(repr(_units()).encode())
    return h.hexdigest()


# ---- evidence guard: which sources a measured kernel was built from
# -----------------------------------
# bench.py quotes counter passes committed under profiles/ (pmc_*.json).  Each such entry records the hash below of the
# kernel's translation unit AS IT WAS WHEN THE COUNTERS WERE COLLECTED; bench.py prints `lookup_stale` and
# tests/test_lookup_evidence.py fails when a committed entry no longer matches the tree.
KERNEL_UNITS = {"rollout_onchip_kernel": "rollout_integrator.hip", "rollout_cost_kernel": "rollout_integrator.hip",
                "rollout_kmppi_kernel": "rollout_integrator.hip", "rollout_mlp_split_kernel": "rollout_mlp_split.hip",
                "rollout_mlp_mfma_kernel": "rollout_mlp_mfma.hip", "weights_partial_rows_kernel": "noise_torch.hip",
                "weights_partial_diag_kernel": "update.hip", "noise_fill_philox_kernel": "update.hip",
                        "finalize_blocks_kernel": "update.hip"}


def _include_closure(src, seen=None):
    """`src` (a file of csrc/) and every csrc/ or include/ header it reaches through #include "..." """
    import re
    seen = [] if seen is None else seen
    path = os.path.join(CSRC, src) if os.path.exists(os.path.join(CSRC, src)) else os.path.join(INCLUDE, src)
    if not os.path.exists(path) or path in seen:
        return seen
    seen.append(path)
    for inc in re.findall(r'^\s*#\s*include\s+"([^"]+)"', open(path).read(), flags=re.M):
        _include_closure(os.path.basename(inc), seen)
    return seen


def kernel_sources_hash(kernel):
    """sha256 over the translation unit that instantiates `kernel` (KERNEL_UNITS), every header it includes, and its
    flags"""
    unit = KERNEL_UNITS[kernel]
    h = hashlib.sha256()
    for path in sorted(_include_closure(unit)):
        h.update(os.path.basename(path).encode())
        h.update(open(path, "rb").read())
    h.update(" ".join(FLAGS + EXTRA.get(unit, [])).encode())
    return h.hexdigest()


def is_current():
    stamp = LIB + ".stamp"
    return os.path.exists(LIB) and os.path.exists(stamp) and open(stamp).read().strip() == _deps_hash()


def build(force=False, verbose=True):
    """Compile every HIP translation unit for gfx950 and link libmppi_amd.so."""
    if not force and is_current():
        return LIB
    hipcc = _hipcc()
    os.makedirs(OBJ_DIR, exist_ok=True)
    srcs = [u for u in _units() if os.path.exists(os.path.join(CSRC, u[0]))]

    # what every unit sees besides its own source: the headers of csrc/ and the C-ABI header
    hh = hashlib.sha256()
    for n in sorted(os.listdir(CSRC)):
        if n.endswith((".hpp", ".h")):
            hh.update(n.encode())
            hh.update(open(os.path.join(CSRC, n), "rb").read())
    hh.update(open(os.path.join(INCLUDE, "mppi_amd.h"), "rb").read())
    headers = hh.hexdigest()

    def one(unit):
        src, objname, defs = unit
        obj = os.path.join(OBJ_DIR, objname)
        cmd = [hipcc, *FLAGS, *EXTRA.get(src, []), *defs, "-I", INCLUDE, "-c", os.path.join(CSRC, src), "-o", obj]
        # an object is kept while its source, the headers and its command line are what they were (one edited .hip = one
        # unit)
        key = hashlib.sha256((headers + " ".join(cmd[1:])).encode() + open(os.path.join(CSRC, src),
                "rb").read()).hexdigest()
        stamp = obj + ".stamp"
        if not force and os.path.exists(obj) and os.path.exists(stamp) and open(stamp).read().strip() == key:
            return obj
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{r.stdout}\n{r.stderr}")
        with open(stamp, "w") as f:
            f.write(key)
        return obj

    with cf.ThreadPoolExecutor(max_workers=min(os.cpu_count() or 8, len(srcs))) as ex:
        objs = list(ex.map(one, srcs))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs, "-ldl"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    with open(LIB + ".stamp", "w") as f:
        f.write(_deps_hash())
    if verbose:
        print(f"[pytorch_mppi_amd] built {LIB}", file=sys.stderr)
    return LIB


# ---- host AddressSanitizer variant (CPU only -- the pool refuses GPU ASan)
# ----------
# The C-ABI's own host code (argument validation, workspace carving, the RCCL binding, the device group's threads and
# hand-over protocol: capi.hip, dist.hip, group.hip) compiled with -fsanitize=address on the HOST side only
# (-Xarch_host: the device code is the product's) and linked with the product build's other objects into
# libmppi_amd_asan.so.  tests/test_abi_asan.py drives the
# refusal paths of every entry point through it from a sanitized C client.
ASAN_UNITS = ["capi.hip", "dist.hip", "group.hip"]
ASAN_LIB = os.path.join(HERE, "libmppi_amd_asan.so")


def asan_runtime_dir():
    """directory of clang's shared ASan runtime (libclang_rt.asan-x86_64.so), or None"""
    import glob
    for base in ("/opt/rocm/lib/llvm/lib/clang", "/opt/rocm/llvm/lib/clang"):
        for p in sorted(glob.glob(os.path.join(base, "*", "lib", "linux", "libclang_rt.asan-x86_64.so"))):
            return os.path.dirname(p)
    return None


def build_asan(verbose=False):
    """libmppi_amd_asan.so (see above); needs the product build's objects (build() runs first).  Returns the path."""
    build(verbose=verbose)
    hipcc = _hipcc()
    d = os.path.join(CSRC, "build_asan")
    os.makedirs(d, exist_ok=True)
    key = hashlib.sha256((_deps_hash() + "asan-1").encode()).hexdigest()
    stamp = ASAN_LIB + ".stamp"
    if os.path.exists(ASAN_LIB) and os.path.exists(stamp) and open(stamp).read().strip() == key:
        return ASAN_LIB
    san = ["-O1", "-g", "-Xarch_host", "-fsanitize=address", "-Xarch_host", "-fno-omit-frame-pointer"]

    def one(src):
        obj = os.path.join(d, src.replace(".hip", ".o"))
        cmd = [hipcc, *[f for f in FLAGS if f != "-O3"], *san, "-I", INCLUDE, "-c", os.path.join(CSRC, src), "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc (ASan host build) failed on {src}:\n{r.stdout}\n{r.stderr}")
        return obj

    with cf.ThreadPoolExecutor(max_workers=len(ASAN_UNITS)) as ex:
        objs = list(ex.map(one, ASAN_UNITS))
    skip = {u.replace(".hip", ".o") for u in ASAN_UNITS}
    others = [os.path.join(OBJ_DIR, u[1]) for u in _units() if u[1] not in skip]
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-fsanitize=address", "-shared-libasan", "-o", ASAN_LIB,
            *objs, *others, "-ldl"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link (ASan host build) failed:\n{r.stdout}\n{r.stderr}")
    with open(stamp, "w") as f:
        f.write(key)
    return ASAN_LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
