import os, sys, time, tempfile
import numpy as np
sys.path.insert(0, os.getcwd())
import tts_cpp_amd
from tts_cpp_amd import synth, runner
os.environ["TTS_KOKORO_INPUT_IS_PHONEMES"] = "1"
fmodel = synth.build_kokoro(synth.kokoro_82m(forced_frames=3))
rng = np.random.default_rng(1)
with tempfile.TemporaryDirectory() as td:
    gpath = fmodel.write_gguf(os.path.join(td, "k.gguf"))
    texts = ["".join(chr(0x61 + int(v)) for v in rng.integers(0, 26, 398)) for _ in range(8)]
    for lanes in (1, 2, 4, 8):
        r = runner.Runner(gpath, voice=b"af_test", max_seqs=lanes)
        r.generate_batch_sizes(texts[:max(2, lanes)], voice=b"af_test")
        if lanes == 4: os.environ["TTS_KOKORO_BATCH_TRACE"] = "1"
        t0 = time.perf_counter(); s = r.generate_batch_sizes(texts, voice=b"af_test"); dt = time.perf_counter() - t0
        os.environ.pop("TTS_KOKORO_BATCH_TRACE", None)
        print("lanes", lanes, "ms per utterance", dt * 1e3 / len(texts), "audio-s/s", sum(s) / 24000 / dt, flush=True)
        t0 = time.perf_counter(); a = r.generate(texts[0], voice=b"af_test"); print("  one generate() ms", (time.perf_counter() - t0) * 1e3)
        r.close()
