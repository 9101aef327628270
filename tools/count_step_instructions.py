#!/usr/bin/env python3
"""Instruction census of the split predict kernel's HOT LOOP (the fused k-step below the diagonal block) from the compiler's
assembly: how many matrix, vector, LDS, LDS-DMA and scalar instructions ONE wave issues per k-step -- the evidence behind
DESIGN 4.1's "vector instructions per MFMA" (SQ_INSTS_VALU counts the MFMAs too).

    hipcc --offload-arch=gfx950 -O3 ... --cuda-device-only -S predict_split_f32.hip -o K.s
    python tools/count_step_instructions.py K.s [kernel-name-substring]

The hot loop = the INNERMOST loop (a backward branch around no other) with the most v_mfma instructions inside the kernel."""
import re
import sys
from collections import Counter

path = sys.argv[1]
want = sys.argv[2] if len(sys.argv) > 2 else "leaf_tiles_bf16_kernelILi2EfLi0ELb1ELb1ELi1E"
lines = open(path).read().splitlines()
start = next(i for i, l in enumerate(lines) if l.startswith("_ZN") and want in l and l.rstrip().endswith(":") or (want in l and l.startswith("_ZN") and ": " in l and l.split(":")[0].startswith("_ZN")))
end = next(i for i in range(start + 1, len(lines)) if lines[i].startswith("\t.section") or lines[i].startswith(".Lfunc_end"))
body = lines[start:end]
labels = {m.group(1): i for i, l in enumerate(body) if (m := re.match(r"^(\.LBB\d+_\d+):", l))}


def klass(op):
    if op.startswith("v_mfma"):
        return "mfma"
    if op.startswith(("v_exp", "v_sqrt", "v_rcp", "v_rsq", "v_log", "v_sin", "v_cos")):
        return "valu_transcendental"
    if op.startswith("v_"):
        return "valu"
    if op.startswith(("ds_read", "ds_load")):
        return "lds_read"
    if op.startswith(("ds_write", "ds_store")):
        return "lds_write"
    if "lds" in op and op.startswith(("global_load", "buffer_load")):
        return "lds_dma"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith("s_waitcnt"):
        return "s_waitcnt"
    if op.startswith("s_barrier"):
        return "s_barrier"
    if op.startswith("s_"):
        return "salu"
    return "other"


loops = []
for i, l in enumerate(body):
    m = re.match(r"^\s+s_cbranch_\w+\s+(\.LBB\d+_\d+)", l) or re.match(r"^\s+s_branch\s+(\.LBB\d+_\d+)", l)
    if m and m.group(1) in labels and labels[m.group(1)] < i:
        seg = body[labels[m.group(1)]:i + 1]
        ops = [t.split()[0] for t in (x.strip() for x in seg) if t and not t.startswith((";", ".", "//")) and not t.endswith(":")]
        # (block placement can put the accumulators' zero-initialisation -- a run of register moves executed once -- between
        # the loop's label and its back edge: runs of >= 16 consecutive moves are not part of an iteration)
        keep, run = [], []
        for o in ops + ["end"]:
            if o.startswith("v_mov_b"):
                run.append(o)
                continue
            if len(run) < 16:
                keep.extend(run)
            run = []
            keep.append(o)
        ops = keep[:-1]
        c = Counter(klass(o) for o in ops)
        loops.append((c["mfma"], labels[m.group(1)], i, c, ops))
# the hot loop = the INNERMOST loop with the most MFMAs: since round 6 the kernel's outer loop walks the row blocks (its body holds
# the diagonal block's unrolled steps too); an innermost loop is one that contains no other loop's back edge
inner = [t for t in loops if not any(o is not t and t[1] <= o[1] and o[2] <= t[2] and (o[1], o[2]) != (t[1], t[2]) for o in loops)]
loops = sorted(inner, reverse=True, key=lambda t: t[0] / max(1, t[3]['s_barrier']))  # (most MFMAs per k-step: the steps below the diagonal block)
total = Counter(klass(t.split()[0]) for t in (x.strip() for x in body) if t and not t.startswith((";", ".", "//")) and not t.endswith(":"))
print(f"kernel: {len(body)} lines; whole-kernel census: {dict(total)}")
for nm, a, b, c, ops in loops[:3]:
    vec = c["valu"] + c["valu_transcendental"]
    print(f"loop lines {a}..{b}: {dict(c)}")
    steps = max(1, c["s_barrier"])  # one workgroup barrier per k-step: the compiler unrolls the loop over the ring of three input buffers
    c = Counter({k: v / steps for k, v in c.items()})
    nm = c["mfma"]
    vec = c["valu"] + c["valu_transcendental"]
    print(f"   = {steps} k-steps per iteration")
    print(f"   per k-step and wave: {nm} MFMA {vec} vector instructions besides "
          f"({c['valu_transcendental']} transcendental), {c['lds_read']} LDS reads, {c['lds_dma']} LDS-DMA  ->  {vec / max(nm, 1):.2f} vector per MFMA, "
          f"{(vec + nm) / max(nm, 1):.2f} counting the MFMAs as SQ_INSTS_VALU does")
    # vector-issue-port budget (MI355X_MICROARCH.md, cycle constants): MFMA 8 of its 16 clocks, plain VALU 4, transcendental 8, ds_read ~4
    port = nm * 8 + c["valu"] * 4 + c["valu_transcendental"] * 8 + (c["lds_read"] + c["lds_write"]) * 4
    print(f"   vector issue port: {port} clocks per wave and k-step against {nm * 16} clocks of matrix pipe; two waves per SIMD: {2 * port} against {2 * nm * 16}")
