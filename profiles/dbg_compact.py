import os, sys
import numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/oracle')
import tts_cpp_amd
from tts_cpp_amd import gguf, hip, synth
cfg = synth.small(weight_type=gguf.F16, ctx=80, max_gen=80)
model = synth.build(cfg)
rng = np.random.default_rng(11)
n, cap = 200, 80
lens = rng.integers(4, 61, n)
prompts = [rng.integers(3, cfg.prompt_vocab, int(l)).astype(np.uint32) for l in lens]
n_steps = int(cap - lens.min())
uni = rng.random((n_steps, n, cfg.n_out), dtype=np.float32)
res = []
for compact, pen in (("1", 1.1), ("0", 1.1), ("1", 1.0), ("0", 1.0), ("0", 1.1)):
    os.environ["TTS_HIP_GEN_COMPACT"] = compact
    eng = hip.HipEngine(cfg, max_seqs=n, kv_positions=cap, flags=hip.FLAG_NO_DAC)
    eng.load(model)
    eng.prefill_batch(prompts)
    toks, done = eng.generate_sampled(lens, n_steps, uni, top_k=20, temperature=0.9, repetition_penalty=pen)
    res.append((toks, done)); eng.close()
def cmp(a, b, name):
    (ta, da), (tb, db) = a, b
    bad = []
    for u in range(n):
        k = int(da[u]) if da[u] else n_steps
        d = np.nonzero((ta[:k, u] != tb[:k, u]).any(axis=1))[0]
        if d.size: bad.append((u, int(d[0]), k))
    print(name, 'done equal', np.array_equal(da, db), 'mismatching utterances', len(bad), bad[:10])
cmp(res[0], res[1], 'pen 1.1 compact vs not')
cmp(res[2], res[3], 'pen 1.0 compact vs not')
cmp(res[1], res[4], 'pen 1.1 not vs not (determinism)')
