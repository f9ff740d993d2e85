#!/usr/bin/env python3
"""Scan a gfx950 .s file (hipcc -save-temps) for kernels whose global loads are separated by full waits — `global_load ... s_waitcnt vmcnt(0)
... global_load` chains outside loops are dependent memory round trips that straight-line code would issue together (the batch-1 chain's
LayerNorm prologue and PRO_ATTN prologue were found this way).

  python profiles/tools/isa_serial_loads.py tts_hip-hip-amdgcn-amd-amdhsa-gfx950.s [min_chain]
"""
import re, sys

path = sys.argv[1]
min_chain = int(sys.argv[2]) if len(sys.argv) > 2 else 3
txt = open(path).read()
funcs = re.split(r'\n(?=_Z[\w]+:\s*;? *@)|\n(?=\w+:\s*; @)', txt)
rows = []
for f in funcs:
    m = re.match(r'(\w+):', f)
    if not m or 's_endpgm' not in f: continue
    name = m.group(1)
    chain = best = 0
    pending_load = False
    for ln in f.split('\n'):
        t = ln.strip()
        if t.startswith('global_load') or t.startswith('buffer_load'):
            if pending_load is None:   # a full wait since the last load
                chain += 1
                best = max(best, chain)
            elif not pending_load:
                chain = 1
            pending_load = True
        elif t.startswith('s_waitcnt') and 'vmcnt(0)' in t:
            if pending_load: pending_load = None
        elif t.startswith('s_barrier') or t.startswith('v_mfma'):
            chain = 0; pending_load = False
    if best >= min_chain: rows.append((best, name))
for best, name in sorted(rows, reverse=True):
    print(f"{best:4d} dependent load groups  {name}")

# Second pattern (round 4, gemv_stream_kernel): a PARTIAL wait in the prologue — loads issued, `s_waitcnt vmcnt(k > 0)`, more loads, all before
# the first barrier / MFMA / loop back-edge.  A load under a predicate whose result is copied on the way out of its block (a phi) produces it:
# the copy waits for the first load, and everything the kernel requests afterwards is a second round trip.  Inside pipelined loops the same
# sequence is the double buffering at work, so only the straight-line head of the kernel is scanned.
if len(sys.argv) > 3 and sys.argv[3] == "partial":
    for f in funcs:
        m = re.match(r'(\w+):', f)
        if not m or 's_endpgm' not in f: continue
        loads = waited = hits = 0
        labels = set()
        for ln in f.split('\n'):
            t = ln.strip()
            lm = re.match(r'(\.LBB\w+):', t)
            if lm: labels.add(lm.group(1))
            if t.startswith('global_load') or t.startswith('buffer_load'):
                if waited: hits += 1; waited = 0
                loads += 1
            elif t.startswith('s_waitcnt') and re.search(r'vmcnt\((\d+)\)', t):
                if loads and int(re.search(r'vmcnt\((\d+)\)', t).group(1)) > 0: waited = 1
            elif t.startswith('s_barrier') or t.startswith('v_mfma'):
                break
            elif t.startswith('s_cbranch') and t.split()[-1] in labels:   # a back-edge: the loops start here
                break
        if hits: print(f"{hits:4d} partial waits in the prologue  {m.group(1)}")
