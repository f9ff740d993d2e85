#!/usr/bin/env python3
"""Per-loop instruction census of one kernel's gfx950 assembly (hipcc -S --cuda-device-only): usage isa_loops.py file.s [kernel-substring]"""
import re, sys
text = open(sys.argv[1]).read().split('\n')
sub = sys.argv[2] if len(sys.argv) > 2 else None
if sub:
    s = next(i for i, l in enumerate(text) if re.match(r'^_Z\w*' + re.escape(sub) + r'\w*:', l))
    e = next(i for i in range(s, len(text)) if text[i].startswith('.Lfunc_end'))
    text = text[s:e]
labels = {}
for i, l in enumerate(text):
    m = re.match(r'^(\.LBB\d+_\d+):', l)
    if m: labels[m.group(1)] = i
VALU = re.compile(r'^\s+v_'); SALU = re.compile(r'^\s+s_')
def census(body):
    c = lambda p: sum(1 for x in body if re.search(p, x))
    return dict(mfma=c('v_mfma'), ds_read=c('ds_read'), ds_write=c('ds_write'), gload=c('global_load|buffer_load'), gstore=c('global_store'),
                valu=sum(1 for x in body if VALU.match(x)) - c('v_mfma'), salu=sum(1 for x in body if SALU.match(x)), barrier=c('s_barrier'),
                waitcnt=c('s_waitcnt'), snop=c('s_nop'), scratch=c('scratch_'), lines=len(body))
print("whole:", census(text))
for i, l in enumerate(text):
    m = re.search(r's_c?branch\w* (\.LBB\d+_\d+)', l)
    if m and labels.get(m.group(1), 1 << 30) < i:
        a = labels[m.group(1)]
        print(f"loop {m.group(1)} [{a}-{i}]:", census(text[a:i]))
