"""Host AddressSanitizer pass over the C-ABI's refusal paths.

capi.hip / dist.hip / group.hip are rebuilt with -fsanitize=address on the host side (pytorch_mppi_amd/_build.build_asan) and a
sanitized plain-C client drives every entry point with the bad calls tests/test_abi.py makes from Python -- null blocks, empty and
inconsistent problems, short workspaces and model blobs, bad dtypes, out-of-range ids, the device group's hand-over protocol out of
order, a group whose workers cannot initialise (no GPU here: the create path's cleanup -- threads joined, slots freed -- runs under
the sanitizer).  Every call must be refused with its error code and the sanitizer must stay silent.  CPU only: the pool refuses GPU
ASan; skipped where clang's shared ASan runtime is missing."""
import os
import shutil
import subprocess

import pytest

from pytorch_mppi_amd import _build

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CLIENT = r'''
#include <stdio.h>
#include <string.h>
#include <stdlib.h>
#include "mppi_amd.h"
static int failures = 0;
#define REFUSED(call, want) do { int rc_ = (call); if (rc_ != (want)) { printf("UNEXPECTED %s -> %d (want %d): %s\n", #call, rc_, (want), mppi_last_error()); ++failures; } } while (0)
int main(void) {
  MppiProblem p;
  memset(&p, 0, sizeof p);
  float* buf = (float*)malloc(sizeof(float) * 70000);     /* host memory standing in for device addresses: never dereferenced on the host */
  REFUSED(mppi_rollout_cost(NULL, NULL), MPPI_E_BADARG);
  REFUSED(mppi_command(NULL, 1, NULL), MPPI_E_BADARG);
  REFUSED(mppi_prepare(&p, NULL), MPPI_E_BADARG);                       /* bad dims */
  if (mppi_workspace_elems(&p) != 0 || mppi_workspace_elems(NULL) != 0 || mppi_onchip_spill_elems(NULL) != 0) { printf("UNEXPECTED sizes\n"); ++failures; }
  p.K = 256; p.T = 8; p.nx = 16; p.nu = 4; p.dtype = MPPI_F32; p.lambda_ = 1.0;
  REFUSED(mppi_rollout_cost(&p, NULL), MPPI_E_BADARG);                  /* missing parameter arrays */
  p.U = p.u_init = p.noise_mu = p.noise_L = p.sigma_inv = p.u_min = p.u_max = buf;
  REFUSED(mppi_weights_partial(&p, NULL), MPPI_E_WORKSPACE);            /* no workspace */
  p.workspace = buf; p.workspace_elems = 16;
  REFUSED(mppi_finalize(&p, 1, NULL), MPPI_E_WORKSPACE);                /* too small */
  p.workspace_elems = 70000;
  p.philox_rounds = 9;
  REFUSED(mppi_prepare(&p, NULL), MPPI_E_BADARG);
  p.philox_rounds = 0; p.lambda_ = 0.0;
  REFUSED(mppi_prepare(&p, NULL), MPPI_E_BADARG);
  p.lambda_ = 1.0; p.n_sampler_rows = 2;
  REFUSED(mppi_prepare(&p, NULL), MPPI_E_BADARG);                       /* sampler rows without actions */
  p.n_sampler_rows = 0; p.noise_pitch = 8;
  REFUSED(mppi_prepare(&p, NULL), MPPI_E_BADARG);                       /* pitch < K */
  p.noise_pitch = 0; p.model_id = MPPI_MODEL_MLP; p.hidden = 64; p.model_params = buf; p.model_params_elems = 100;
  REFUSED(mppi_prepare(&p, NULL), MPPI_E_BADARG);                       /* short model blob (ABI 22) */
  p.model_id = MPPI_MODEL_NONE; p.model_params = NULL; p.dtype = 9;
  REFUSED(mppi_combine(&p, buf, 2, NULL), MPPI_E_BADARG);               /* bad dtype */
  p.dtype = MPPI_F64;
  REFUSED(mppi_combine(&p, NULL, 2, NULL), MPPI_E_BADARG);
  const void* ptrs[2] = {buf, NULL};
  p.U_out = buf; p.cost_total = buf;
  REFUSED(mppi_combine_ptrs(&p, ptrs, 2, NULL), MPPI_E_BADARG);         /* a null record pointer */
  REFUSED(mppi_combine_ptrs(&p, ptrs, MPPI_MAX_GROUP + 1, NULL), MPPI_E_BADARG);
  REFUSED(mppi_noise_fill_philox(&p, NULL, NULL), MPPI_E_BADARG);
  REFUSED(mppi_noise_from_ktn(&p, NULL, buf, NULL), MPPI_E_BADARG);
  REFUSED(mppi_kmppi_interp(&p, buf, NULL), MPPI_E_BADARG);             /* no S / theta / W */
  REFUSED(mppi_command_kmppi(&p, NULL, 1, NULL), MPPI_E_BADARG);
  REFUSED(mppi_kmppi_shift(MPPI_F32, 0, 4, 2, buf, buf, buf, buf, buf, buf, NULL), MPPI_E_BADARG);
  REFUSED(mppi_kmppi_trajectory(7, 4, 4, 2, buf, buf, buf, NULL), MPPI_E_BADARG);
  REFUSED(mppi_kmppi_after_update(MPPI_F32, 4, 4, 2, buf, buf, buf, buf, buf, NULL, buf, NULL), MPPI_E_BADARG);
  REFUSED(mppi_smppi_shift(MPPI_F32, 4, 2, NULL, buf, buf, 0.1, buf, buf, buf, NULL), MPPI_E_BADARG);
  REFUSED(mppi_upload_small(buf, 6, buf, NULL), MPPI_E_BADARG);         /* not a multiple of 4 */
  REFUSED(mppi_upload_small(buf, 4096, buf, NULL), MPPI_E_BADARG);
  REFUSED(mppi_register_model(MPPI_MODEL_CUSTOM_BASE + 64, 2, 2, buf, NULL), MPPI_E_BADARG);
  REFUSED(mppi_register_model(MPPI_MODEL_CUSTOM_BASE, 2, 2, NULL, NULL), MPPI_E_BADARG);
  if (mppi_model_supported(MPPI_MODEL_CUSTOM_BASE + 999, 2, 2, MPPI_F32, 0) != 0 || mppi_model_supported(MPPI_MODEL_PENDULUM, 2, 1, 5, 0) != 0) { printf("UNEXPECTED support\n"); ++failures; }
  if (mppi_noise_rows4(0, 4) != 0 || mppi_noise_rows4(64, 12) != 192 || mppi_noise_pitch(0, 0) != 0) { printf("UNEXPECTED geometry\n"); ++failures; }
  /* RCCL binding */
  REFUSED(mppi_dist_unique_id(NULL), MPPI_E_BADARG);
  void* comm = NULL;
  char id[128];
  memset(id, 0, sizeof id);
  REFUSED(mppi_dist_init(id, 3, 2, &comm), MPPI_E_BADARG);              /* rank >= world */
  REFUSED(mppi_dist_init(NULL, 0, 1, &comm), MPPI_E_BADARG);
  REFUSED(mppi_exchange_combine(&p, NULL, buf, 2, NULL), MPPI_E_BADARG);
  REFUSED(mppi_dist_init_all(0, NULL, NULL), MPPI_E_BADARG);
  int32_t devs[3] = {0, 0, 0};
  void* comms[3] = {buf, buf, buf};
  REFUSED(mppi_dist_init_all(2, devs, comms), MPPI_E_UNSUPPORTED);      /* a device listed twice */
  REFUSED(mppi_exchange_combine_all(2, devs, NULL, comms, comms, comms), MPPI_E_BADARG);
  if (mppi_dist_destroy(NULL) != 0) { printf("UNEXPECTED destroy\n"); ++failures; }
  /* the device group */
  void* grp = NULL;
  REFUSED(mppi_group_create(0, devs, NULL, &grp), MPPI_E_BADARG);
  REFUSED(mppi_group_create(MPPI_MAX_GROUP + 1, devs, NULL, &grp), MPPI_E_BADARG);
  REFUSED(mppi_group_create(2, devs, comms, &grp), MPPI_E_UNSUPPORTED);
  REFUSED(mppi_group_submit(NULL, 0, &p, NULL, buf, NULL), MPPI_E_BADARG);
  REFUSED(mppi_group_wait(NULL, NULL, NULL), MPPI_E_BADARG);
  REFUSED(mppi_group_abort(NULL), MPPI_E_BADARG);
  REFUSED(mppi_group_broadcast(NULL, buf, 4, comms, NULL), MPPI_E_BADARG);
  if (mppi_group_destroy(NULL) != 0 || mppi_group_size(NULL) != 0) { printf("UNEXPECTED group\n"); ++failures; }
  /* three workers that cannot make a device current (no GPU): creation fails, every thread is joined, every slot freed */
  int rc = mppi_group_create(3, devs, NULL, &grp);
  if (rc == 0) {
    /* (a GPU is present after all: exercise the protocol errors on the live group instead) */
    REFUSED(mppi_group_wait(grp, NULL, NULL), MPPI_E_BADARG);           /* nothing submitted */
    REFUSED(mppi_group_submit(grp, 7, &p, NULL, buf, NULL), MPPI_E_BADARG);
    REFUSED(mppi_group_submit(grp, 0, &p, NULL, buf, NULL), MPPI_E_BADARG);   /* no record */
    if (mppi_group_abort(grp) != 0 || mppi_group_destroy(grp) != 0) { printf("UNEXPECTED live group\n"); ++failures; }
  } else if (grp != NULL) { printf("UNEXPECTED: failed create left a handle\n"); ++failures; }
  if (mppi_abi_version() != MPPI_ABI_VERSION || mppi_problem_size() != (long long)sizeof(MppiProblem)) { printf("UNEXPECTED abi\n"); ++failures; }
  free(buf);
  printf(failures ? "FAILURES %d\n" : "ALL_REFUSED %d\n", failures);
  return failures ? 1 : 0;
}
'''


def test_refusal_paths_of_the_c_abi_under_host_asan(tmp_path):
    rt = _build.asan_runtime_dir()
    clang = "/opt/rocm/lib/llvm/bin/clang"
    if rt is None or not os.path.exists(clang) or shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"):
        pytest.skip("clang's shared AddressSanitizer runtime (libclang_rt.asan-x86_64.so) is not in this toolchain")
    lib = _build.build_asan()
    src = tmp_path / "client.c"
    src.write_text(CLIENT)
    exe = tmp_path / "client"
    r = subprocess.run([clang, "-std=c99", "-g", "-fsanitize=address", "-shared-libasan", "-fno-omit-frame-pointer",
                        "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe), lib, f"-Wl,-rpath,{os.path.dirname(lib)}",
                        f"-Wl,-rpath,{rt}"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:halt_on_error=1:abort_on_error=0",
               LD_LIBRARY_PATH=os.pathsep.join([rt, os.path.dirname(lib), os.environ.get("LD_LIBRARY_PATH", "")]))
    run = subprocess.run([str(exe)], capture_output=True, text=True, env=env, timeout=120)
    assert "AddressSanitizer" not in run.stderr, run.stderr[-3000:]
    assert run.returncode == 0 and "ALL_REFUSED 0" in run.stdout, (run.returncode, run.stdout[-2000:], run.stderr[-2000:])
