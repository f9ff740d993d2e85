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


def test_isa_scanner_finds_full_and_partial_waits_between_loads(tmp_path):
    """profiles/tools/isa_serial_loads.py on hand-written gfx950 assembly: (1) `global_load ... s_waitcnt vmcnt(0) ... global_load` chains count as
    dependent load groups; (2) mode `partial`: loads, `s_waitcnt vmcnt(k > 0)`, more loads in a kernel's straight-line head is flagged (the shape
    gemv_stream_kernel's conditional first weight load had until round 4), the same sequence behind a loop back-edge, a barrier or an MFMA is not."""
    import subprocess, sys
    asm = tmp_path / "k.s"
    asm.write_text("""
chain_kernel: ; @chain_kernel
	global_load_dword v1, v[2:3], off
	s_waitcnt vmcnt(0)
	global_load_dword v4, v[1:2], off
	s_waitcnt vmcnt(0)
	global_load_dword v5, v[4:5], off
	s_waitcnt vmcnt(0)
	s_endpgm
head_kernel: ; @head_kernel
	global_load_dwordx4 v[16:19], v[2:3], off
	global_load_dwordx4 v[20:23], v[2:3], off offset:64
	s_waitcnt vmcnt(1)
	v_mov_b32_e32 v9, v16
	global_load_dwordx4 v[24:27], v[4:5], off
	s_waitcnt vmcnt(0)
	s_barrier
	s_endpgm
loop_kernel: ; @loop_kernel
	global_load_dwordx4 v[16:19], v[2:3], off
.LBB2_1:
	global_load_dwordx4 v[20:23], v[2:3], off offset:64
	s_waitcnt vmcnt(1)
	v_mfma_f32_16x16x32_f16 a[0:3], v[16:19], v[8:11], a[0:3]
	global_load_dwordx4 v[16:19], v[2:3], off offset:128
	s_cbranch_scc1 .LBB2_1
	s_endpgm
""")
    tool = os.path.join(ROOT, "profiles", "tools", "isa_serial_loads.py")
    full = subprocess.run([sys.executable, tool, str(asm), "2"], capture_output=True, text=True, check=True).stdout
    assert "3 dependent load groups  chain_kernel" in full and "head_kernel" not in full and "loop_kernel" not in full, full
    part = subprocess.run([sys.executable, tool, str(asm), "99", "partial"], capture_output=True, text=True, check=True).stdout
    assert "1 partial waits in the prologue  head_kernel" in part and "loop_kernel" not in part and "chain_kernel" not in part, part


def test_bench_time_budget_keeps_the_contract_and_drops_extras_in_order():
    """bench.py's wall-clock budget with the costs measured in round 5 (`time_budget.sections` of profiles/r05/bench_driver_flags.json): under the
    driver's round-end flags (25 steps of ~11.3 s already spent) every extra section fits the default 550 s; a tighter budget drops the long parts from
    the end of their order (uniform_same_mix, then ragged, then ragged_stream, then uniform) and keeps the cheap sections; `--time-budget-s 0` keeps all."""
    b = _bench()
    extras = ["decode_step_batch1", "generate_batch1_end_to_end", "secondary.kokoro", "secondary.dia", "secondary.orpheus",
              "long_utterances.uniform", "long_utterances.ragged_stream", "long_utterances.ragged", "long_utterances.uniform_same_mix"]

    def run(spent, budget_s):
        tb, now = b.TimeBudget(budget_s), [float(spent)]
        tb.elapsed = lambda: now[0]
        tb.reserved = b.TimeBudget.RESERVED          # the CPU baseline still to come while the first two extras are decided
        ran = []
        for i, sec in enumerate(extras):
            if i == 2:
                now[0] += b.TimeBudget.RESERVED
                tb.reserved = 0.0
            if tb.room(sec):
                ran.append(sec)
                now[0] += tb.COST[sec]
        return now[0], ran, tb.skipped

    spent = 5 + 25 * 11.3 + 8                                     # imports + setup, 25 steps, the roofline pass (294 s in the measured line)
    total, ran, skipped = run(spent, 550)
    assert total <= 550 and ran == extras and skipped == []
    total, ran, skipped = run(spent, 480)
    assert total <= 480 and ran == extras[:7] and skipped == extras[7:]
    total, ran, skipped = run(spent, 350)
    assert total <= 350 and ran == extras[:5] and skipped == extras[5:]
    total, ran, skipped = run(spent, 0)
    assert ran == extras and skipped == []
    rep = b.TimeBudget(480).report()
    assert rep["budget_s"] == 480 and rep["skipped"] == [] and rep["elapsed_s"] >= 0 and rep["sections"] == {}



def test_wait_order_scanner_flags_weights_retired_before_the_prologue(tmp_path):
    """profiles/tools/isa_wait_order.py (round 5): a wait that retires a streamed (non-temporal) load while staging loads still follow is reported;
    the same kernel with the staging loads requested first is not; --first-use ends the head at a prologue's own arithmetic."""
    asm = tmp_path / "k.s"
    asm.write_text("""
weights_first_kernel:                   ; @weights_first_kernel
\tglobal_load_dwordx4 v[2:5], v[0:1], off nt
\tglobal_load_dword v6, v[0:1], off
\ts_waitcnt vmcnt(0)
\tglobal_load_dword v7, v[0:1], off
\ts_waitcnt vmcnt(0)
\tds_write_b32 v8, v7
\ts_barrier
\tv_mfma_f32_16x16x32_f16 v[0:3], v[2:5], v[2:5], v[0:3]
\ts_endpgm
inputs_first_kernel:                    ; @inputs_first_kernel
\tglobal_load_dword v6, v[0:1], off
\tglobal_load_dword v7, v[0:1], off
\tglobal_load_dwordx4 v[2:5], v[0:1], off nt
\ts_waitcnt vmcnt(1)
\tds_write_b32 v8, v7
\ts_barrier
\ts_waitcnt vmcnt(0)
\tv_mfma_f32_16x16x32_f16 v[0:3], v[2:5], v[2:5], v[0:3]
\ts_endpgm
softmax_prologue_kernel:                ; @softmax_prologue_kernel
\tglobal_load_dword v6, v[0:1], off
\tglobal_load_dwordx4 v[2:5], v[0:1], off nt
\ts_waitcnt vmcnt(1)
\tv_exp_f32_e32 v9, v6
\ts_waitcnt vmcnt(0)
\tglobal_load_dword v7, v[0:1], off
\ts_waitcnt vmcnt(0)
\tds_write_b32 v8, v7
\ts_barrier
\tv_mfma_f32_16x16x32_f16 v[0:3], v[2:5], v[2:5], v[0:3]
\ts_endpgm
""")
    tool = os.path.join(ROOT, "profiles", "tools", "isa_wait_order.py")
    out = subprocess.run([sys.executable, tool, str(asm)], capture_output=True, text=True, check=True).stdout
    assert "weights_first_kernel" in out and "inputs_first_kernel" not in out
    assert "softmax_prologue_kernel" in out            # its later chunk of keys does wait for the weights ...
    out = subprocess.run([sys.executable, tool, str(asm), "softmax_prologue_kernel", "--first-use=v_exp_f32"], capture_output=True, text=True, check=True).stdout
    assert "softmax_prologue_kernel" not in out        # ... but its first arithmetic starts under them


def test_trace_steps_tool_splits_a_kernel_trace_into_steps(tmp_path):
    """profiles/tools/trace_steps.py (round 5): the dispatches behind the N-th last launch of the marker kernel, per kernel and per step."""
    rows = ["Kernel_Name,Start_Timestamp,End_Timestamp"]
    t = 1000
    for step in range(6):
        for name, dur in (("dia_embed_kernel(DiaEmbedArgs)", 5000), ("void gemv_stream_kernel<4, 0, 0, 8>(GemmArgs, StreamMap)", 10000), ("rms_fold_rows_kernel(float*)", 4000),
                          ("void gemv_stream_kernel<4, 0, 0, 8>(GemmArgs, StreamMap)", 10000)):
            rows.append(f'"{name}",{t},{t + dur}')
            t += dur + 2000
    csvf = tmp_path / "kernel_trace.csv"
    csvf.write_text("\n".join(rows) + "\n")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "profiles", "tools", "trace_steps.py"), str(csvf), "dia_embed_kernel", "4"],
                         capture_output=True, text=True, check=True).stdout
    head = out.splitlines()[0]
    assert head.startswith("4 steps, 37.0 us per step, 29.0 us in kernels, 4 launches per step"), head
    line = [ln for ln in out.splitlines() if "gemv_stream_kernel<4, 0, 0, 8>" in ln][0].split()
    assert "2.0" in line and "10.00" in line and "20.0" in line, line
