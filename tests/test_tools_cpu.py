"""Host-side measurement tooling (no GPU): how bench.py prices a kernel family, and the ISA scanner that found the serialized epilogues."""
import importlib.util
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_codec_families_are_priced_on_the_pipe_they_run_on():
    """bf16 x 3 families: issued flops (6 x algorithmic) against the dense bf16 peak, only while tts_hip_dac_arith() says so; the k = 1 convs of
    the wide classes are bf16 x 3 only with split planes (bit 5) — the committed line of call 33 still priced them on the fp32 pipe at 0.82."""
    b = _bench()
    st = dict(ms_total=10.0, launches=10, bytes_total=1e9, flops_total=1.0e12)   # 100 TFLOP/s algorithmic
    k1 = next(n for n, keys in b.FAMILIES.items() if keys == ["dac_conv1"])
    k7 = next(n for n, keys in b.FAMILIES.items() if keys == ["dac_conv7"])
    on = b.roof_of(k1, st, "t", 10.0, 7 | 32, "f32")
    assert on["peak"] == b.F16_PEAK_TFLOPS and abs(on["achieved"] - 600.0) < 1e-6 and abs(on["fp32_equivalent_TFLOPs"] - 100.0) < 1e-6
    off = b.roof_of(k1, st, "t", 10.0, 7, "f32")             # planes off: exact-fp32 MFMA
    assert off["peak"] == b.F32_PEAK_TFLOPS and abs(off["achieved"] - 100.0) < 1e-6
    assert b.roof_of(k7, st, "t", 10.0, 1, "f32")["peak"] == b.F16_PEAK_TFLOPS
    assert b.roof_of(k7, st, "t", 10.0, 0, "f32")["peak"] == b.F32_PEAK_TFLOPS       # round-2 arithmetic
    assert b.roof_of(k7, st, "t", 10.0, 8, "f16")["peak"] == b.F16_PEAK_TFLOPS       # F16 tensors: plain fp16 MFMA, algorithmic flops
    attn = next(n for n, keys in b.FAMILIES.items() if keys == ["attn_self"])
    r = b.roof_of(attn, st, "t", 10.0, 39, "f32")
    assert r["bound"] == "hbm" and r["peak"] == b.HBM_PEAK_GBS and abs(r["achieved"] - 100.0) < 1e-6


def test_isa_scanner_counts_full_waits_between_loads(tmp_path):
    """profiles/tools/isa_serial_loads.py: a load issued after a `s_waitcnt vmcnt(0)` that followed another load is a dependent round trip."""
    asm = tmp_path / "k.s"
    asm.write_text("""
serial_kernel:                          ; @serial_kernel
\tglobal_load_dword v1, v[2:3], off
\ts_waitcnt vmcnt(0)
\tglobal_load_dword v4, v[2:3], off
\ts_waitcnt vmcnt(0)
\tglobal_load_dword v5, v[2:3], off
\ts_waitcnt vmcnt(0)
\tglobal_store_dword v[2:3], v5, off
\ts_endpgm
batched_kernel:                         ; @batched_kernel
\tglobal_load_dword v1, v[2:3], off
\tglobal_load_dword v4, v[2:3], off
\tglobal_load_dword v5, v[2:3], off
\ts_waitcnt vmcnt(0)
\tglobal_store_dword v[2:3], v5, off
\ts_endpgm
""")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "profiles", "tools", "isa_serial_loads.py"), str(asm), "2"],
                         capture_output=True, text=True, check=True).stdout
    assert "serial_kernel" in out and "batched_kernel" not in out
    assert out.strip().split()[0] == "3"
