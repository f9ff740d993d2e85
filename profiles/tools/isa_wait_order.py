#!/usr/bin/env python3
"""Which prologue waits retire the weight stream?  (round 5)

vmcnt counts in issue order: a wave that has requested its streamed weights (the non-temporal loads) and THEN loads its small inputs
(activation rows, norm weights, positions: L2 hits) cannot use the small inputs before the weights have landed — the wait for an input
issued later retires everything issued earlier.  A loop header whose first iteration meets pending loads gets a conservative `vmcnt(0)`
from the compiler with the same effect.  Either way the staging prologue the weights were meant to fly under starts one HBM round trip late.

For every kernel: walk the straight-line head (up to the first integer dot / MFMA / v_dot2 or the end), keep the queue of outstanding
loads, and report each `s_waitcnt vmcnt(N)` that retires a non-temporal load while work that is NOT the consumer of those weights follows:
further global loads, LDS writes or a barrier before the first dot product.

  python profiles/tools/isa_wait_order.py unit.s [name-filter] [--loops] [--first-use=MNEMONIC]
      --loops: walk into the loops of the head too
      --first-use=v_exp_f32: the head ends at this instruction as well (a prologue whose own arithmetic is what has to start before the weights land:
                             the attention inside gemm16_kernel<.., PRO_CROSS, ..>, whose later chunks of keys, E > 8, do wait for the weights)
"""
import re, sys

txt = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 and not sys.argv[2].startswith("--") else ""
funcs = re.split(r'\n(?=_Z[\w]+:\s*;? *@)|\n(?=\w+:\s*; @)', txt)
DOT = ("v_dot4", "v_dot2", "v_mfma", "v_dot8") + tuple(a.split("=", 1)[1] for a in sys.argv if a.startswith("--first-use="))
for f in funcs:
    m = re.match(r'(\w+):', f)
    if not m or 's_endpgm' not in f or flt not in m.group(1):
        continue
    lines = [l.strip() for l in f.split('\n')]
    queue = []          # outstanding loads: (line index, is_nt)
    events = []         # (line index, n_nt_retired, n_other_retired)
    first_dot = None
    skip_to = None      # inside a loop of the head (remainder passes: zero trips at the shapes these kernels run at, reported as a count)
    loops = 0
    for i, t in enumerate(lines):
        lm = re.match(r'(\.LBB\w+):.*Loop Header', t)
        if skip_to is None and lm and "--loops" not in sys.argv:
            skip_to = lm.group(1); loops += 1
            continue
        if skip_to is not None:
            if t.startswith("s_cbranch") and t.split()[-1] == skip_to:
                skip_to = None
            continue
        if t.startswith(("global_load", "buffer_load")):
            queue.append((i, " nt" in t or "slc" in t))
        elif t.startswith("s_waitcnt"):
            mm = re.search(r'vmcnt\((\d+)\)', t)
            if mm:
                keep = int(mm.group(1))
                gone = queue[:max(0, len(queue) - keep)]
                queue = queue[max(0, len(queue) - keep):]
                nt = sum(1 for g in gone if g[1])
                if nt:
                    events.append((i, nt, len(gone) - nt))
        elif t.startswith(DOT):
            first_dot = i
            break
    if not events:
        continue
    end = first_dot if first_dot is not None else len(lines)
    for (i, nt, other) in events:
        after, sk = [], None
        for t in lines[i + 1:end]:
            lm = re.match(r'(\.LBB\w+):.*Loop Header', t)
            if sk is None and lm and "--loops" not in sys.argv: sk = lm.group(1); continue
            if sk is not None:
                if t.startswith("s_cbranch") and t.split()[-1] == sk: sk = None
                continue
            after.append(t)
        loads = sum(1 for t in after if t.startswith(("global_load", "buffer_load")))
        bars = sum(1 for t in after if t.startswith("s_barrier"))
        lds = sum(1 for t in after if t.startswith("ds_write"))
        if loads or bars:
            print(f"{m.group(1)[:100]}: line {i}: wait retires {nt} streamed loads (+{other} others); before the first dot product still to come: "
                  f"{loads} global loads, {bars} barriers, {lds} LDS writes ({loops} loops of the head skipped)")
            break
