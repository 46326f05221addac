"""CPU: the C-ABI library builds for gfx950, loads without a GPU and exports every symbol that
include/mppi_amd.h declares; host-side argument validation refuses bad calls before touching HIP."""
import ctypes as C
import os
import re

import pytest

from pytorch_mppi_amd import _native as N

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "mppi_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mppi_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_all_exported_and_bound():
    lib = N.lib()
    declared = _declared_symbols()
    assert len(declared) >= 14
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/mppi_amd.h but not exported"
        assert name in N.SYMBOLS, f"{name} has no ctypes prototype in _native.SYMBOLS"
    assert sorted(N.SYMBOLS) == declared


def test_abi_version_and_struct_mirror():
    lib = N.lib()
    assert lib.mppi_abi_version() == N.ABI_VERSION
    assert lib.mppi_problem_size() == C.sizeof(N.MppiProblem)
    hdr = open(os.path.join(ROOT, "include", "mppi_amd.h")).read()
    assert f"#define MPPI_ABI_VERSION {N.ABI_VERSION}" in hdr
    # field order of the mirror == field order of the header
    body = hdr[hdr.index("typedef struct MppiProblem {") + len("typedef struct MppiProblem {"):hdr.index("} MppiProblem;")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    names = []
    for decl in body.split(";"):
        decl = decl.strip()
        if not decl or decl.startswith("typedef"):
            continue
        parts = decl.replace("*", " ").split(",")
        first = parts[0].split()
        names.append(first[-1])
        names += [x.strip() for x in parts[1:]]
    assert names == [f[0] for f in N.MppiProblem._fields_]


@pytest.mark.parametrize("T,nu,rows", [(64, 12, 192), (15, 1, 4), (10, 2, 5), (10, 3, 9), (32, 1, 8), (8, 4, 8), (7, 6, 12)])
def test_noise_rows4(T, nu, rows):
    assert N.noise_rows4(T, nu) == rows
    assert rows * 4 >= T * nu


@pytest.mark.parametrize("K,dtype,pitch", [(65536, N.F32, 65536), (131072, N.F32, 131072 + 65536), (262144, N.F32, 262144 + 65536),
                                           (100000, N.F32, 100000), (8192, N.F64, 8192), (65536, N.F64, 65536 + 32768),
                                           (393216, N.F32, 393216 + 65536)])
def test_noise_row_pitch(K, dtype, pitch):
    """rows that would be a multiple of 2 MiB are padded by 1 MiB (HBM bank aliasing between streamed rows)"""
    assert N.noise_pitch(K, dtype) == pitch


def _spill_rows(nu, T, nta=5):
    """where the rows of a sample wait in the on-chip command (csrc/common.hpp onchip_geometry), restated: registers hold the first
    nta weighting tiles, LDS whole tiles as far as 160 KB go beside the tables, the spill array the rest -- in rows-of-4"""
    g = 4 if nu % 4 == 0 else (2 if nu % 2 == 0 else 1)
    p4, tt = nu // g, 4 // g
    sw = max(16 // p4, 1)
    rg = 1 if p4 >= 3 else (2 if p4 == 2 else 4)
    ag = nta * sw
    nss = -(-T // tt)
    ntiles = -(-nss // sw)
    tables = 3 * nss * p4 * 4 + 4
    base = (tables + 4 * ntiles * 64) * 4
    if p4 > 16 or sw % rg or base > 160 * 1024:
        return 0
    room = (160 * 1024 - base) // (p4 * 256 * 16)
    nsl = max(0, min(nss - ag, room))
    nsl -= nsl % sw
    return max(0, ntiles * sw - ag - nsl) * p4


@pytest.mark.parametrize("nu,T", [(12, 64), (12, 20), (12, 200), (4, 100), (4, 30), (6, 40), (2, 130), (1, 48), (3, 33), (8, 256), (16, 64)])
def test_onchip_spill_size_is_the_geometry_restated(nu, T):
    """ABI 20 `mppi_onchip_spill_elems`: rows-of-4 that fit neither registers nor LDS x padded samples x 4; 0 for fp64"""
    for K in (65536, 1000):
        p = N.MppiProblem()
        p.K, p.T, p.nx, p.nu, p.dtype, p.sigma_diagonal = K, T, 4, nu, N.F32, 1
        kpad = -(-K // 256) * 256
        assert int(N.lib().mppi_onchip_spill_elems(C.byref(p))) == _spill_rows(nu, T) * kpad * 4, (nu, T, K)
        p.dtype = N.F64
        assert int(N.lib().mppi_onchip_spill_elems(C.byref(p))) == 0
    if (nu, T) == (12, 64):
        assert _spill_rows(nu, T) == 90           # C3: 25 super-steps in registers, 10 in LDS, 6 tiles = 30 super-steps x 3 rows wait


def _pair_spill_rows(nu, nx, T, plain=True, kt=1, pbrows=12):
    """the same for the two-waves-per-sample form (csrc/common.hpp onchip_pair_geometry), restated: rows-of-4 per THREAD of the
    512-thread workgroup; 0 when the form does not apply"""
    g = 4 if nu % 4 == 0 else (2 if nu % 2 == 0 else 1)
    p4, tt = nu // g, 4 // g
    sw = max(16 // p4, 1)
    ch = max(pbrows // p4, 1)
    if not plain and ch >= 2:
        ch -= 1                       # the form with the SMPPI terms compiled in: chunks one super-step shorter
    kr = kt * sw
    nss = -(-T // tt)
    nch = -(-nss // ch)
    nit = (nch + 1) // 2
    nls = nit * ch
    ntl = -(-nls // sw)
    shn = max((nx + 2 + (0 if plain else nu)) * 256, 8 * ntl * 64)
    fixed = (3 * nss * p4 * 4 + 16 + shn) * 4
    if p4 > 16 or fixed > 160 * 1024 or nch < 4:
        return 0
    room = (160 * 1024 - fixed) // (p4 * 512 * 16)
    nsl = max(0, min((nch // 2) * ch - kr, room))
    nsl -= nsl % sw
    after = ntl * sw - kr - nsl                       # local super-steps beyond registers and LDS, padding included
    if after <= 0:
        return 0

    def real(ls):
        return ls < nls and (2 * (ls // ch)) * ch + ls % ch < nss
    areal = sum(1 for ls in range(kr + nsl, ntl * sw) if real(ls))
    best, dls = 1e30, 0
    for c in range(0, 4):                              # whole tiles generated a second time: the share closest to a third
        if c * sw > after:
            break
        d = abs(sum(1 for ls in range((ntl - c) * sw, ntl * sw) if real(ls)) - 0.32 * areal)
        if d < best - 1e-9:
            best, dls = d, c * sw
    return (after - dls) * p4


@pytest.mark.parametrize("T", [64, 48, 33, 100, 15, 200])
def test_onchip_spill_size_covers_the_two_wave_form(T):
    """round 6: the array is the larger of the two forms' needs for the model the pair kernel is instantiated for (integrator
    16 x 12); at C3 (T = 64) the two coincide: 90 rows x 256 = 45 rows x 512 per workgroup"""
    p = N.MppiProblem()
    p.K, p.T, p.nx, p.nu, p.dtype, p.sigma_diagonal, p.model_id = 65536, T, 16, 12, N.F32, 1, N.MODEL_INTEGRATOR
    one = _spill_rows(12, T) * 65536 * 4
    two = max(_pair_spill_rows(12, 16, T, plain) for plain in (True, False)) * 2 * 65536 * 4
    # (a horizon whose rows all fit registers + LDS of the one-wave kernel has no array, and without one the pair kernel is not used)
    assert int(N.lib().mppi_onchip_spill_elems(C.byref(p))) == (max(one, two) if one else 0), (T, one, two)
    if T == 64:
        assert one == _pair_spill_rows(12, 16, T) * 2 * 65536 * 4 == 90 * 65536 * 4
    p.model_id = N.MODEL_LINEAR_GOAL
    assert int(N.lib().mppi_onchip_spill_elems(C.byref(p))) == one


def test_model_support_table():
    assert N.model_supported(N.MODEL_PENDULUM, 2, 1, N.F32)
    assert N.model_supported(N.MODEL_PENDULUM, 2, 1, N.F64)
    assert N.model_supported(N.MODEL_INTEGRATOR, 16, 12, N.F32)
    assert N.model_supported(N.MODEL_LINEAR_GOAL, 2, 2, N.F64)
    assert N.model_supported(N.MODEL_MLP, 16, 4, N.F32, 256)
    assert not N.model_supported(N.MODEL_MLP, 16, 4, N.F32, 0)
    assert not N.model_supported(N.MODEL_INTEGRATOR, 7, 5, N.F32)
    assert not N.model_supported(N.MODEL_NONE, 2, 2, N.F32)
    assert not N.model_supported(N.MODEL_PENDULUM, 2, 1, 7)


def test_bad_calls_refused_on_host():
    lib = N.lib()
    assert lib.mppi_rollout_cost(None, None) == -1
    p = N.MppiProblem()
    assert lib.mppi_workspace_elems(C.byref(p)) == 0
    p.K, p.T, p.nx, p.nu, p.dtype = 256, 8, 2, 2, N.F32
    need = lib.mppi_workspace_elems(C.byref(p))
    assert need > 0
    p.lambda_ = 1.0
    assert lib.mppi_rollout_cost(C.byref(p), None) == -1          # missing parameter arrays
    assert b"missing" in lib.mppi_last_error()
    p.dtype = 9
    assert lib.mppi_prepare(C.byref(p), None) == -1
    assert b"dtype" in lib.mppi_last_error()
    with pytest.raises(RuntimeError, match="code -1"):
        N.check(-1, "x")


def test_a_short_model_blob_is_refused_not_read_past():
    """ABI 22: the MLP blob grew by qx | qu in ABI 21; a caller still passing the older, shorter layout used to get
    garbage cost weights.  `model_params_elems` is checked before anything is launched (fake non-null addresses: nothing here is
    dereferenced on the host)."""
    lib = N.lib()
    import numpy as np
    keep = np.zeros(1 << 16, dtype=np.float32)
    addr = keep.ctypes.data
    p = N.MppiProblem()
    p.K, p.T, p.nx, p.nu, p.dtype, p.hidden, p.model_id, p.lambda_ = 256, 8, 16, 4, N.F32, 64, N.MODEL_MLP, 1.0
    for f in ("U", "u_init", "noise_mu", "noise_L", "sigma_inv", "u_min", "u_max", "model_params", "workspace"):
        setattr(p, f, addr)
    p.workspace_elems = keep.size
    need = 64 * 20 + 64 + 16 * 64 + 2 * 16 + 1 + 4
    p.model_params_elems = need - 20                               # the ABI-20 blob: no qx | qu behind the residual scale
    assert lib.mppi_prepare(C.byref(p), None) == -1
    msg = lib.mppi_last_error().decode()
    assert "model_params holds" in msg and str(need) in msg, msg
    p.model_params_elems = 0                                       # unstated counts as too short, too
    assert lib.mppi_prepare(C.byref(p), None) == -1 and b"model_params holds" in lib.mppi_last_error()
    p.model_params_elems = need
    rc = lib.mppi_prepare(C.byref(p), None)                        # past that check: refused for the missing noise stream instead
    assert rc == -1 and b"model_params holds" not in lib.mppi_last_error()


def test_device_group_entry_points_refuse_bad_calls_on_the_host():
    """csrc/group.hip (ABI 22): argument checks need no GPU; creating a group without one fails cleanly (the workers cannot make
    their device current) instead of hanging"""
    lib = N.lib()
    grp = C.c_void_p()
    devs = (C.c_int32 * 2)(0, 0)
    assert lib.mppi_group_create(0, devs, None, C.byref(grp)) == -1
    assert lib.mppi_group_create(2, None, None, C.byref(grp)) == -1
    comms = (C.c_void_p * 2)(1, 1)
    assert lib.mppi_group_create(2, devs, comms, C.byref(grp)) == N.E_UNSUPPORTED      # RCCL: one rank per device
    assert b"listed twice" in lib.mppi_last_error()
    assert lib.mppi_group_wait(None, None, None) == -1 and lib.mppi_group_abort(None) == -1
    assert lib.mppi_group_submit(None, 0, None, None, None, None) == -1
    assert lib.mppi_group_destroy(None) == 0 and lib.mppi_group_size(None) == 0
    import torch
    if not torch.cuda.is_available():
        rc = lib.mppi_group_create(2, devs, None, C.byref(grp))
        assert rc != 0 and not grp.value, "no GPU: the workers cannot initialise, the group must not exist"


def test_plain_c_client_links_and_agrees_on_the_struct(tmp_path):
    """The boundary is a C ABI: the header compiles as strict C99 (no C++, no torch types), a C program links
    against the shared library, sees the same ABI version and sizeof(MppiProblem), and a compute entry point
    refuses an empty problem on the host (no GPU needed)."""
    import shutil
    import subprocess
    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("no gcc")
    src = tmp_path / "client.c"
    src.write_text(
        '#include <stdio.h>\n#include <string.h>\n#include "mppi_amd.h"\n'
        "int main(void) {\n"
        "  MppiProblem p; memset(&p, 0, sizeof p);\n"
        '  printf("%d %lld %lld %d %d\\n", mppi_abi_version(), (long long)mppi_problem_size(), (long long)sizeof p,\n'
        "         mppi_rollout_cost(&p, NULL), mppi_rollout_cost_kmppi(&p, NULL));\n"
        "  return 0;\n}\n")
    exe = tmp_path / "client"
    from pytorch_mppi_amd import _build
    N.lib()                                    # builds the library if it is not there yet
    libdir = os.path.dirname(_build.LIB)
    r = subprocess.run([gcc, "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.join(ROOT, "include"), str(src),
                        "-o", str(exe), "-L", libdir, "-l:" + os.path.basename(_build.LIB), "-Wl,-rpath," + libdir,
                        "-Wl,--allow-shlib-undefined"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    ver, size_lib, size_c, rc1, rc2 = (int(x) for x in out.stdout.split())
    assert ver == N.ABI_VERSION and size_lib == size_c == C.sizeof(N.MppiProblem)
    assert rc1 == -1 and rc2 == -1             # MPPI_E_BADARG


def test_integration_md_stub_mirrors_the_struct():
    """the ctypes stub INTEGRATION.md shows a reference maintainer is the struct the library was compiled with"""
    src = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "INTEGRATION.md")).read()
    i, j = src.index("class MppiProblem(C.Structure):"), src.index("assert _lib.mppi_abi_version()")
    ns = {}
    exec("import ctypes as C\n" + src[i:j], ns)
    stub = ns["MppiProblem"]
    assert C.sizeof(stub) == int(N.lib().mppi_problem_size())
    assert [(n, t) for n, t in stub._fields_] == [(n, t) for n, t in N.MppiProblem._fields_]
    assert f"mppi_abi_version() == {N.ABI_VERSION}" in src
