"""Register / scratch / LDS use of every kernel in a built library, read from its gfx950 code objects.

`kernels(path)` copies the shared object into a temporary directory, extracts the offload bundles (`llvm-objdump
--offloading` writes them next to its input), reads the AMDGPU metadata note of each gfx950 code object (`llvm-readelf
--notes`) and returns {demangled kernel name: {vgpr, agpr, sgpr, scratch, vgpr_spill, sgpr_spill, lds}}.  Used by
tests/test_kernel_resources.py (a kernel that starts spilling is a performance regression no parity test sees: VERDICT
r03 weak #2, the KMPPI-fused K1 went from 20 B to 528 B of scratch and from 70.7 to 94 us unnoticed) and by
tools/kernel_resources.py.  Needs no GPU."""
import os
import re
import shutil
import subprocess
import tempfile

_LLVM = "/opt/rocm/lib/llvm/bin"
_FIELDS = {".vgpr_count": "vgpr", ".agpr_count": "agpr", ".sgpr_count": "sgpr",
        ".private_segment_fixed_size": "scratch",
           ".vgpr_spill_count": "vgpr_spill", ".sgpr_spill_count": "sgpr_spill", ".group_segment_fixed_size": "lds"}


def _tool(name):
    for c in (os.path.join(_LLVM, name), shutil.which(name)):
        if c and os.path.exists(c):
            return c
    raise RuntimeError(f"{name} not found (ROCm's llvm tools)")


def kernels(path, demangle=True):
    out = {}
    with tempfile.TemporaryDirectory() as d:
        lib = os.path.join(d, os.path.basename(path))
        shutil.copy(path, lib)
        subprocess.run([_tool("llvm-objdump"), "--offloading", lib], cwd=d, capture_output=True, check=True)
        for f in sorted(os.listdir(d)):
            if "gfx950" not in f:
                continue
            notes = subprocess.run([_tool("llvm-readelf"), "--notes", os.path.join(d, f)], capture_output=True,
                    text=True, check=True).stdout
            cur = None
            for line in notes.splitlines():
                s = line.strip()
                if s.startswith("- ."):                     # first key of a new kernel record (or of an .args entry)
                    if line.startswith("  - "):
                        cur = {}
                    s = s[2:]
                if cur is None or ":" not in s:
                    continue
                k, v = s.split(":", 1)
                if k in _FIELDS and line.startswith("    ") and not line.startswith("      "):
                    cur[_FIELDS[k]] = int(v)
                elif k == ".name" and line.startswith("    ") and not line.startswith("      "):
                    cur["_name"] = v.strip()
                elif k == ".wavefront_size" and "_name" in cur:
                    out[cur.pop("_name")] = cur
                    cur = None
    if demangle and out:
        filt = shutil.which("c++filt") or _tool("llvm-cxxfilt")
        names = list(out)
        dem = subprocess.run([filt], input="\n".join(names), capture_output=True, text=True,
                check=True).stdout.splitlines()
        out = {re.sub(r"\s+", " ", dn): out[n] for n, dn in zip(names, dem)}
    return out


def find(table, *needles):
    """the kernels whose demangled name contains every needle"""
    return {k: v for k, v in table.items() if all(n in k for n in needles)}
