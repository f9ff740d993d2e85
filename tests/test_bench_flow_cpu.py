"""bench.py's main() end to end on a CPU box: the device side (runners, the HIP library, torch.cuda, the secondary engines) replaced by stand-ins that
advance a fake clock by what each piece costs on an MI355X.  What is checked is the control flow the driver depends on: one JSON line with the
contract's fields whatever the flags, and the wall-clock budget of the extra sections (`--time-budget-s`) under the driver's round-end flags."""
import contextlib
import importlib.util
import io
import json
import os
import sys
import types

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture()
def bench(monkeypatch):
    import torch
    spec = importlib.util.spec_from_file_location("bench_flow_mod", os.path.join(ROOT, "bench.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    clock = [15.0]   # imports and start-up
    b.time = types.SimpleNamespace(perf_counter=lambda: clock[0])
    b._T_START = 0.0

    class FakeRunner:
        def __init__(self, *a, **k): pass
        def device_context(self): return 1
        def tokenize(self, t): return [1, 2, 3]
        def generate_batch_sizes(self, texts):
            clock[0] += 11.0
            return [1000] * len(texts)
        def close(self): pass

    class FakeLib:
        def __getattr__(self, name):
            return lambda *a, **k: {"tts_hip_last_error": b"", "tts_hip_arena_bytes": 1248 << 20, "tts_hip_dac_arith": 39}.get(name, 0)

    class FakeModel:
        kv = []
        def write_gguf(self, p): open(p, "w").close()

    def run_all(runners, texts, timings_=None, stream=False, reps=1):   # reps: the timed region's runners make their K calls back to back (round 6)
        n = sum(len(t) for t in texts)
        warm = len(texts[0]) > 1 and texts[0][0] == texts[0][-1] and texts[0][0].startswith("w" * 40)   # the long section's 48-step warm-up
        clock[0] += reps * ((61.0 if n > 100 else 2.0) if stream else (11.8 if len(texts[0]) >= 1000 else 5.0 if warm else 33.0))
        if timings_ is not None:
            timings_.append(9.7)
        return reps * n * 100000

    def costs(seconds, ret):
        def f(*a, **k):
            clock[0] += seconds
            return ret
        return f

    stats = {k: dict(ms_total=1.0, launches=10, bytes_total=1e9, flops_total=1e9) for keys in b.FAMILIES.values() for k in keys}
    monkeypatch.setattr(b.hip, "load_lib", lambda: FakeLib())
    monkeypatch.setattr(b.synth, "build", lambda cfg, shapes_only=False: FakeModel())
    monkeypatch.setattr(b.runner, "Runner", FakeRunner)
    b.load_runners = lambda *a, **k: ([FakeRunner() for _ in range(3)], None)
    b.make_sentences = lambda rn, n, plen, seed: ["x"] * n
    b.long_sentences = lambda rn, n, lo, hi, seed: ["w" * 50] if n == 1 else ["x" * (1 + i % 7) for i in range(n)]
    b.run_all = run_all
    b.profile_get = lambda L, rn: stats
    b.pmc_traffic = lambda *a, **k: None
    b.decode_step_sweep = costs(5, {"steps_1024": {}})
    b.generate_batch1_end_to_end = costs(8, {"top_k_50": {}})
    b.cpu_baseline = costs(14, {"value": 0.1, "unit": "audio-seconds/sec", "cores": 32, "kind": "port", "sample": "-"})
    monkeypatch.setattr(torch.cuda, "set_device", lambda d: None)
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 1)
    monkeypatch.setattr(torch.cuda, "synchronize", lambda *a: None)
    monkeypatch.setattr(torch.cuda, "mem_get_info", lambda d=0: (280 << 30, 288 << 30))
    sec = types.ModuleType("secondary_bench")
    sec.RUNNERS = {"kokoro": costs(2, {"value": 598.0}), "dia": costs(8, {"value": 20.0}), "orpheus": costs(5, {"value": 9.4})}
    monkeypatch.setitem(sys.modules, "secondary_bench", sec)

    def go(argv, start=15.0):
        clock[0] = start
        monkeypatch.setattr(sys, "argv", ["bench.py"] + argv)
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            b.main()
        lines = [ln for ln in buf.getvalue().strip().split("\n") if ln.startswith("{")]
        assert len(lines) == 1                                  # ONE JSON line
        return json.loads(lines[0]), clock[0]
    return go


CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
            "roofline", "cpu_baseline")


def test_driver_flags_keep_the_contract_and_stay_inside_the_budget(bench):
    """the driver's round-end flags with this round's measured costs (profiles/r05/bench_full_call3.json: time_budget.sections): every extra
    section fits the default budget — round 4's line dropped the two ragged parts of long_utterances — and the whole run stays under it"""
    d, total = bench(["--gpus", "1", "--steps", "20", "--warmup", "5"])
    assert all(k in d for k in CONTRACT) and d["steps"] == 20 and d["warmup"] == 5 and d["n_gpus"] == 1
    assert d["roofline"]["bound"] in ("hbm", "mfma") and "frac" in d["roofline"] and "traffic" in d["roofline"]
    assert total <= 550 and d["time_budget"]["skipped"] == [], (total, d["time_budget"])
    lu = d["long_utterances"]
    assert all("audio_seconds_per_sec" in lu[k] for k in ("uniform", "ragged_stream", "ragged", "uniform_same_mix"))
    assert "of_uniform_same_mix" in lu["ragged_stream"] and "of_uniform_same_mix" in lu["ragged"] and "of_lockstep_ragged" in lu["ragged_stream"]
    assert all("value" in v for v in d["secondary"].values()) and "decode_step_batch1" in d and "generate_batch1_end_to_end" in d
    assert set(d["time_budget"]["sections"]) >= {"decode_step_batch1", "cpu_baseline", "long_utterances.uniform", "long_utterances.ragged"}
    # a slow first `import torch` eats into the extras (last wanted first), never into the contract
    d, total = bench(["--gpus", "1", "--steps", "20", "--warmup", "5"], start=100.0)
    assert all(k in d for k in CONTRACT) and total <= 550 + 10
    assert d["time_budget"]["skipped"] and d["time_budget"]["skipped"][-1] == "long_utterances.uniform_same_mix" and "audio_seconds_per_sec" in d["long_utterances"]["uniform"]


def test_default_and_unlimited_runs(bench):
    d, _ = bench([])
    assert all(k in d for k in CONTRACT) and d["steps"] == 3 and d["warmup"] == 1
    assert {"uniform", "ragged", "ragged_stream", "uniform_same_mix"} <= set(d["long_utterances"]) and "of_uniform" in d["long_utterances"]["ragged_stream"]
    d, _ = bench(["--time-budget-s", "0", "--steps", "20", "--warmup", "5"])
    assert d["time_budget"]["skipped"] == [] and "of_lockstep_ragged" in d["long_utterances"]["ragged_stream"]
    d, _ = bench(["--no-long", "--no-secondary", "--no-e2e", "--no-step-sweep"])
    assert all(k in d for k in CONTRACT) and "long_utterances" not in d and "secondary" not in d
