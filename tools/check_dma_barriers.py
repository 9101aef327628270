#!/usr/bin/env python3
"""Static check of the compiler's assembly: can a wave reach an s_barrier with LDS-DMA (global_load_lds / buffer_load ... lds) of
its own still in flight?  A workgroup barrier that publishes LDS-DMA data must be preceded -- on EVERY path -- by a wait for the
wave's DMAs (s_waitcnt vmcnt).  hipcc 7.2 inserts that wait when the DMA and the barrier share a basic block or a loop
iteration, and was found (round 6) NOT to insert it when the DMAs are issued before a loop's back edge and the barrier stands at
the loop's header.  This tool does the data-flow over the kernel's control-flow graph:

    state = the number of the wave's LDS-DMAs that may be outstanding (saturating at 64), joined by max over predecessors
    global_load_lds* / buffer_load*lds : state += 1      s_waitcnt vmcnt(N) : state = min(state, N)      s_barrier : report state

    hipcc --offload-arch=gfx950 -O3 ... --cuda-device-only -S file.hip -o file.s
    python tools/check_dma_barriers.py [--allow=kernel_name_substring ...] file.s [...]
                                                             exit status 1 if any barrier may be reached with DMAs pending

A kernel that keeps DMAs in flight across a barrier ON PURPOSE (a deeper ring: the barrier publishes an older DMA, waited for with
vmcnt(N > 0)) shows up with its N; the report lists every barrier with a non-zero state so that each can be read."""
import re
import sys
from collections import defaultdict

SAT = 64


def kernels(lines):
    start = None
    for i, l in enumerate(lines):
        if re.match(r"^_Z\w+:", l) or re.match(r"^[A-Za-z_]\w*:\s*; @", l):
            start = (i, l.split(":")[0])
        elif l.startswith(".Lfunc_end") and start:
            yield start[1], lines[start[0] + 1:i]
            start = None


def analyse(name, body):
    # basic blocks
    blocks, cur, label = [], [], "entry"
    order = []
    for l in body:
        t = l.strip()
        m = re.match(r"^(\.LBB\d+_\d+):", t)
        if m:
            blocks.append((label, cur))
            label, cur = m.group(1), []
            continue
        if not t or t.startswith((";", ".", "//")):
            continue
        cur.append(t)
    blocks.append((label, cur))
    idx = {lab: i for i, (lab, _) in enumerate(blocks)}
    succ = defaultdict(list)
    for i, (lab, ins) in enumerate(blocks):
        fall = True
        for t in ins:
            m = re.match(r"^s_cbranch_\w+\s+(\.LBB\d+_\d+)", t)
            if m and m.group(1) in idx:
                succ[i].append(idx[m.group(1)])
            m = re.match(r"^s_branch\s+(\.LBB\d+_\d+)", t)
            if m and m.group(1) in idx:
                succ[i].append(idx[m.group(1)])
                fall = False
            if t.startswith(("s_endpgm", "s_setpc")):
                fall = False
        if fall and i + 1 < len(blocks):
            succ[i].append(i + 1)

    def transfer(state, ins, report=None, lab=None):
        for k, t in enumerate(ins):
            op = t.split()[0]
            if (op.startswith("global_load_lds") or (op.startswith("buffer_load") and " lds" in t)):
                state = min(SAT, state + 1)
            elif op == "s_waitcnt":
                m = re.search(r"vmcnt\((\d+)\)", t)
                if m:
                    state = min(state, int(m.group(1)))
                elif re.fullmatch(r"s_waitcnt\s+0(x0+)?", t):
                    state = 0
            elif op == "s_barrier" and report is not None and state > 0:
                report.append((lab, k, state))
        return state

    inn = [0] * len(blocks)
    changed = True
    while changed:
        changed = False
        for i, (lab, ins) in enumerate(blocks):
            out = transfer(inn[i], ins)
            for j in succ[i]:
                if out > inn[j]:
                    inn[j] = out
                    changed = True
    report = []
    nb = 0
    ndma = 0
    for i, (lab, ins) in enumerate(blocks):
        nb += sum(1 for t in ins if t.split()[0] == "s_barrier")
        ndma += sum(1 for t in ins if t.split()[0].startswith("global_load_lds") or (t.split()[0].startswith("buffer_load") and " lds" in t))
        transfer(inn[i], ins, report, lab)
    return nb, ndma, report


# kernels that keep DMAs in flight across a barrier on purpose (counted s_waitcnt vmcnt(N) + raw s_barrier): named with --allow
allow = [a.split("=", 1)[1] for a in sys.argv[1:] if a.startswith("--allow=")]
bad = 0
for path in [a for a in sys.argv[1:] if not a.startswith("--")]:
    lines = open(path).read().splitlines()
    for name, body in kernels(lines):
        nb, ndma, report = analyse(name, body)
        if ndma == 0 or nb == 0:
            continue
        short = name if len(name) < 100 else name[:97] + "..."
        if report and any(a in name for a in allow):
            worst = max(r[2] for r in report)
            print(f"by design {short}: {len(report)} of {nb} barriers with up to {worst} LDS-DMAs in flight (a counted wait in front of a raw s_barrier: a ring deeper than two)")
        elif report:
            bad += 1
            worst = max(r[2] for r in report)
            print(f"PENDING  {short}: {len(report)} of {nb} barriers may be reached with up to {worst if worst < SAT else 'many'} LDS-DMAs in flight "
                  f"({ndma} DMA instructions); first: block {report[0][0]}")
        else:
            print(f"ok       {short}: {nb} barriers, {ndma} DMA instructions, every barrier behind a full wait")
sys.exit(1 if bad else 0)
