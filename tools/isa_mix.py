"""Instruction mix of a kernel's hot loop, from the shipped code object (no GPU): VERDICT r03 item 7 asks for an
instruction-level issue budget of the split MLP kernel (rollout_mlp_split_kernel<16,0,true>) -- "tabulate VALU cycles per
activation by class".

    python tools/isa_mix.py <library.so> <demangled-name substring> [out.txt]

Extracts the gfx950 code objects (llvm-objdump --offloading), disassembles the kernel, takes its LONGEST backward-branch loop as
the hot loop and counts instructions by class:
    mfma | trans (v_exp / v_rcp / v_log / v_sqrt / v_rsq / v_sin / v_cos) | cvt+pack (v_cvt_*, v_pack_*) | fma_mix | valu (everything
    else on the vector ALU) | salu | lds | vmem | wait/nop
and prints an ISSUE estimate with the per-class costs measured on this chip at one wave per SIMD (MI355X_MICROARCH.md,
DESIGN.md 3): plain VALU 4 cycles, transcendental 16, 16-bit MFMA 16x16x32 one issue slot of 4 (its 16 pipe cycles overlap the
VALU work between two MFMAs), SALU / waits not counted (they issue beside the vector ALU)."""
import os
import re
import shutil
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
TRANS = ("v_exp_", "v_rcp_", "v_log_", "v_sqrt_", "v_rsq_", "v_sin_", "v_cos_")
COST = {"mfma": 4, "trans": 16, "cvt_pack": 4, "fma_mix": 4, "valu": 4, "dpp_permlane": 4}


def classify(op):
    if op.startswith("v_mfma") or op.startswith("v_smfmac"):
        return "mfma"
    if op.startswith(TRANS):
        return "trans"
    if op.startswith("v_fma_mix"):
        return "fma_mix"
    if op.startswith(("v_cvt_", "v_pack_")):
        return "cvt_pack"
    if op.startswith(("v_permlane", "v_readlane", "v_writelane", "v_readfirstlane")) or "_dpp" in op:
        return "dpp_permlane"
    if op.startswith("v_accvgpr"):
        return "accvgpr"
    if op.startswith("v_"):
        return "valu"
    if op.startswith(("s_waitcnt", "s_nop", "s_barrier", "s_sleep", "s_setprio", "s_sched")):
        return "wait_nop"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    return "other"


def disassemble(lib, needle):
    with tempfile.TemporaryDirectory() as d:
        p = os.path.join(d, os.path.basename(lib))
        shutil.copy(lib, p)
        subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", p], cwd=d, capture_output=True, check=True)
        for f in sorted(os.listdir(d)):
            if "gfx950" not in f:
                continue
            co = os.path.join(d, f)
            syms = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-t", "-C", co], capture_output=True, text=True).stdout
            hit = [l for l in syms.splitlines() if needle in l and " F " in l and ".kd" not in l]
            if not hit:
                continue
            raw = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-t", co], capture_output=True, text=True).stdout
            addr = hit[0].split()[0]
            mangled = [l.split()[-1] for l in raw.splitlines() if l.startswith(addr) and " F " in l][0]
            dis = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", f"--disassemble-symbols={mangled}", co], capture_output=True, text=True).stdout
            return hit[0].split(" F ")[-1].strip(), dis
    raise SystemExit(f"no kernel matching {needle!r} in {lib}")


def main(lib, needle, out=None):
    name, dis = disassemble(lib, needle)
    ins = []      # (address, opcode, operands)
    for l in dis.splitlines():
        m = re.match(r"\s+([a-z_0-9]+)\s*(.*?)\s*//\s*([0-9A-F]+):(.*)$", l)
        if m:
            ins.append((int(m.group(3), 16), m.group(1), m.group(2), m.group(4)))
    addr_index = {a: i for i, (a, _, _, _) in enumerate(ins)}
    loops = []
    for i, (a, op, args, tail) in enumerate(ins):
        if op.startswith("s_cbranch") or op == "s_branch":
            m = re.search(r"<[^>]*\+0x([0-9a-fA-F]+)>", tail)            # branch target as symbol + offset
            tgt = ins[0][0] + int(m.group(1), 16) if m else None
            if tgt is not None and tgt in addr_index and tgt <= a:
                loops.append((addr_index[tgt], i))
    ins = [(a, op, args) for a, op, args, _ in ins]
    lines = [f"# {name}", f"# {len(ins)} instructions; backward branches (loops): {len(loops)}"]
    if not loops:
        lines.append("# no loop found")
    else:
        lo, hi = max(loops, key=lambda ab: ab[1] - ab[0])
        body = ins[lo:hi + 1]
        cnt = {}
        for _, op, _ in body:
            cnt[classify(op)] = cnt.get(classify(op), 0) + 1
        est = sum(COST.get(k, 0) * v for k, v in cnt.items())
        lines.append(f"# hot loop: instructions {lo}..{hi} ({len(body)} instructions per iteration)")
        lines.append(f"{'class':>14} {'count':>7} {'issue cycles (est.)':>20} {'share':>7}")
        for k, v in sorted(cnt.items(), key=lambda kv: -COST.get(kv[0], 0) * kv[1]):
            c = COST.get(k, 0) * v
            lines.append(f"{k:>14} {v:7d} {c:20d} {100.0 * c / est if est else 0:6.1f}%")
        lines.append(f"{'total':>14} {len(body):7d} {est:20d}")
        tops = {}
        for _, op, _ in body:
            tops[op] = tops.get(op, 0) + 1
        lines.append("# most frequent opcodes: " + ", ".join(f"{o} x{n}" for o, n in sorted(tops.items(), key=lambda kv: -kv[1])[:24]))
    txt = "\n".join(lines) + "\n"
    if out:
        open(out, "w").write(txt)
    print(txt)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else None)
