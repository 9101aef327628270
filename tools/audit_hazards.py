#!/usr/bin/env python3
"""
Disassembly audit of libgpso_hip.so (gfx950), run by tests/test_cabi_cpu.py on every build:

  1. NO packed FP32 VALU operation (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 / v_pk_mov_b32 feeding them) in any
     kernel.  The one silently wrong result this engine has produced (profiles/r02h_packed_mean_bug.txt: the high
     half of a dependent v_pk_fma_f32 chain came back wrong, timing dependent, on some boxes) needed a packed chain;
     the library is therefore built with the target feature `packed-fp32-ops` switched OFF (csrc/Makefile), which
     removes the ingredient by construction -- explicit vector arithmetic and __builtin_elementwise_fma included --
     and this audit fails the CPU suite if one ever comes back.
  2. For every v_mfma_f64_16x16x4_f64 (16 passes = 64 clocks on gfx950, twice gfx942's) the number of WAIT STATES
     (instructions issued, `s_nop N` counting N + 1: the unit of the ISA's "manually inserted wait states" tables and
     of LLVM's hazard recogniser) before the first instruction on ANY path -- branches followed, not a linear scan --
     that reads or overwrites one of its destination registers.  The hardware does not interlock this dependency;
     hipcc 7.2 pads it to 19 wait states (measured: `s_nop 15; s_nop 2` between the MFMA and a v_fmac_f64 of its
     result; 18 before a store / ds_write of it) and 10 for the 8-pass v_mfma_f32_16x16x4_f32.  Inline asm is invisible to that pass, so the audit checks
     the result: every distance >= the bound.  Another MFMA that consumes the registers (accumulation, srcC) is a
     dependency inside the matrix pipe with its own table and is counted apart ("chained").

Usage: audit_hazards.py [path/to/libgpso_hip.so] [--json out.json] [--verbose]
Exit status 1 when a rule is violated.
"""
import json
import os
import re
import subprocess
import sys
import tempfile

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
# wait states hipcc 7.2 leaves between an MFMA and the first VALU / memory instruction touching its result (gfx950)
# (VALU consumer, memory / LDS consumer): measured on hipcc 7.2 -- `s_nop 15; s_nop 2` before a v_fmac_f64 of the result,
# `s_nop 15; s_nop 1` before a global_store / ds_write of it; `s_nop 9` either way for the 8-pass f32 MFMA
BOUND = {"v_mfma_f64_16x16x4_f64": (19, 18), "v_mfma_f32_16x16x4_f32": (10, 10)}
MEM_PREFIXES = ("ds_", "global_", "buffer_", "flat_", "scratch_")


def bound_for(mfma_op, consumer):
    valu, mem = BOUND[mfma_op]
    return mem if consumer.startswith(MEM_PREFIXES) else valu
PACKED_F32 = re.compile(r"^v_pk_(fma|mul|add)_f32$")
HORIZON = 40  # wait states beyond which a path is no longer followed


def disassemble(so_path, workdir):
    """-> list of (bundle name, disassembly text) of the gfx950 code objects embedded in the shared library."""
    local = os.path.join(workdir, "lib.so")
    with open(so_path, "rb") as src, open(local, "wb") as dst:
        dst.write(src.read())
    subprocess.run([OBJDUMP, "--offloading", local], check=True, cwd=workdir, stdout=subprocess.DEVNULL)
    out = []
    for name in sorted(os.listdir(workdir)):
        if "gfx950" not in name or os.path.getsize(os.path.join(workdir, name)) == 0:
            continue
        text = subprocess.run([OBJDUMP, "-d", "--mcpu=gfx950", name], check=True, cwd=workdir, capture_output=True,
                              text=True).stdout
        out.append((name, text))
    if not out:
        raise RuntimeError(f"no gfx950 code object found in {so_path}")
    return out


_REG = re.compile(r"(?<![A-Za-z0-9_])([va])(?:\[(\d+):(\d+)\]|(\d+)(?![0-9A-Za-z_]))")


def regs_of(operands):
    """set of ('v' | 'a', index) named in an operand string"""
    found = set()
    for m in _REG.finditer(operands):
        cls = m.group(1)
        if m.group(2) is not None:
            found.update((cls, i) for i in range(int(m.group(2)), int(m.group(3)) + 1))
        else:
            found.add((cls, int(m.group(4))))
    return found


class Insn:
    __slots__ = ("addr", "op", "operands", "target", "regs")

    def __init__(self, addr, op, operands, target):
        self.addr, self.op, self.operands, self.target = addr, op, operands, target
        self.regs = regs_of(operands)


def parse_functions(text):
    """-> {function name: [Insn]} in address order"""
    funcs, cur = {}, None
    head = re.compile(r"^([0-9a-f]+) <(.+)>:$")
    line_re = re.compile(r"^\s+(\S+)\s*(.*?)\s*//\s*([0-9A-Fa-f]+):")
    tgt_re = re.compile(r"<[^>+]+\+0x([0-9a-fA-F]+)>\s*$|<([^>+]+)>\s*$")
    base = 0
    for line in text.splitlines():
        m = head.match(line)
        if m:
            cur = funcs.setdefault(m.group(2), [])
            base = int(m.group(1), 16)
            continue
        if cur is None:
            continue
        m = line_re.match(line)
        if not m:
            continue
        op, operands, addr = m.group(1), m.group(2), int(m.group(3), 16)
        target = None
        if op.startswith("s_cbranch") or op == "s_branch":
            t = tgt_re.search(line)
            if t:
                target = base + (int(t.group(1), 16) if t.group(1) else 0)
        cur.append(Insn(addr, op, operands, target))
    return funcs


def wait_states(insn):
    if insn.op == "s_nop":
        return int(insn.operands.split()[0], 0) + 1
    return 1


def mfma_dst_and_c(insn):
    """destination registers and the registers of srcC (None when srcC is a constant)"""
    parts = [p.strip() for p in insn.operands.split(",")]
    dst = regs_of(parts[0])
    src_c = regs_of(parts[3]) if len(parts) > 3 else set()
    return dst, src_c


def audit_function(insns):
    """-> (packed ops [(addr, op)], mfma records [(addr, op, min distance, first toucher, chained)])"""
    index = {ins.addr: i for i, ins in enumerate(insns)}
    packed = [(ins.addr, ins.op) for ins in insns if PACKED_F32.match(ins.op)]
    records = []
    for i, ins in enumerate(insns):
        if ins.op not in BOUND:
            continue
        dst, _ = mfma_dst_and_c(ins)
        best, toucher, chained = None, None, False
        # walk every path from the instruction behind the MFMA; state = (instruction index, wait states so far)
        seen = {}
        stack = [(i + 1, 0)]
        while stack:
            j, ws = stack.pop()
            while j < len(insns):
                if ws >= HORIZON or seen.get(j, 1 << 30) <= ws:
                    break
                seen[j] = ws
                cur = insns[j]
                if cur.regs & dst:
                    if cur.op.startswith("v_mfma"):
                        # another MFMA consuming / accumulating onto the result: a dependency INSIDE the matrix pipe
                        # (its own, shorter table; srcC of a back-to-back MFMA is forwarded) -- counted, not bounded
                        chained = True
                        break
                    if best is None or ws < best:
                        best, toucher = ws, f"{cur.op} {cur.operands}"
                    break
                ws += wait_states(cur)
                if cur.op == "s_endpgm":
                    break
                if cur.op == "s_branch":
                    j = index.get(cur.target, len(insns))
                    continue
                if cur.op.startswith("s_cbranch") and cur.target in index:
                    stack.append((index[cur.target], ws))
                j += 1
        records.append((ins.addr, ins.op, best, toucher, chained))
    return packed, records


def audit(so_path):
    report = {"library": so_path, "packed_f32": [], "kernels": {}, "violations": []}
    with tempfile.TemporaryDirectory() as tmp:
        for bundle, text in disassemble(so_path, tmp):
            for name, insns in parse_functions(text).items():
                packed, records = audit_function(insns)
                for addr, op in packed:
                    report["packed_f32"].append({"kernel": name, "addr": hex(addr), "op": op})
                    report["violations"].append(f"{name}: {op} at {addr:#x} (packed FP32 operations are banned)")
                if not records:
                    continue
                per_op = {}
                for addr, op, dist, toucher, chained in records:
                    e = per_op.setdefault(op, {"count": 0, "chained": 0, "min_wait_states": None, "at": None, "first_use": None})
                    e["count"] += 1
                    e["chained"] += bool(chained and dist is None)
                    if dist is not None and (e["min_wait_states"] is None or dist < e["min_wait_states"]):
                        e["min_wait_states"], e["at"], e["first_use"] = dist, hex(addr), toucher
                    if dist is not None and dist < bound_for(op, toucher):
                        report["violations"].append(
                            f"{name}: {op} at {addr:#x}: {dist} wait states before `{toucher}` touches its result (< {bound_for(op, toucher)})")
                report["kernels"][name] = per_op
    return report


def main(argv):
    args = [a for a in argv if not a.startswith("--")]
    so = args[0] if args else os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "pygpso_amd", "libgpso_hip.so")
    rep = audit(os.path.abspath(so))
    if "--json" in argv:
        with open(argv[argv.index("--json") + 1], "w") as fh:
            json.dump(rep, fh, indent=1)
    nk = len(rep["kernels"])
    print(f"{rep['library']}: {nk} kernels with f64 / f32 MFMAs audited; packed FP32 operations: {len(rep['packed_f32'])}")
    worst = {}
    for name, per_op in rep["kernels"].items():
        for op, e in per_op.items():
            if e["min_wait_states"] is None:
                continue
            w = worst.get(op)
            if w is None or e["min_wait_states"] < w[0]:
                worst[op] = (e["min_wait_states"], name, e["at"], e["first_use"])
            if "--verbose" in argv:
                print(f"  {op:26s} x{e['count']:4d}  min {e['min_wait_states']:3d} wait states at {e['at']}: {e['first_use']}  [{name[:90]}]")
    for op, (ws, name, at, use) in sorted(worst.items()):
        print(f"  {op}: minimum over the library {ws} wait states (bound {bound_for(op, use)}) at {at} in {name[:100]}: {use}")
    for v in rep["violations"]:
        print("VIOLATION:", v)
    return 1 if rep["violations"] else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
