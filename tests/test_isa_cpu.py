"""Device code checked at the ISA level, without a GPU (hipcc cross-compiles gfx950): the hot kernels of the batch-1 chain and the epilogues
restructured in round 3 must stay free of scratch and of serialized load chains.  Both were found the slow way — a predicate around a load
became its own basic block with an `s_waitcnt vmcnt(0)` behind it (4 dependent round trips in a LayerNorm prologue, 384 in the planes convs'
epilogue, 8 at the end of every tiled decoder GEMM: DESIGN.md §5) — so the scanner that found them now guards them."""
import os
import re
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"

PARLER_TU = """
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include "{root}/tts.cpp_amd/csrc/gemm_tile_kernels.h"
template __global__ void gemm16_kernel<1, PRO_LN, EPI_STORE, 1>(GemmArgs);
template __global__ void gemm16_kernel<1, PRO_LN, EPI_QKV, 1>(GemmArgs);
template __global__ void gemm16_kernel<1, PRO_ATTN, EPI_RESID, 1>(GemmArgs);
template __global__ void gemm16_kernel<1, PRO_F16, EPI_STORE, 1>(GemmArgs);
template __global__ void gemm16_kernel<1, PRO_CROSS, EPI_RESID, 1>(GemmArgs);
template __global__ void gemm_tile_kernel<128, 128, 2, 4, 64, 4, EPI_QKV>(GemmArgs, TileMap);
template __global__ void gemm_tile_kernel<128, 128, 2, 4, 64, 4, EPI_RESID>(GemmArgs, TileMap);
template __global__ void gemm_tile_kernel<64, 64, 2, 4, 128, 3, EPI_RESID>(GemmArgs, TileMap);
template __global__ void attn_rows_kernel<8>(AttnArgs);
template __global__ void ln_rows_t_kernel<4, 0>(float *, int, const float *, const float *, float *, _Float16 *, int, const float *, int, int64_t);
"""
LLAMA_TU = """
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include "{root}/tts.cpp_amd/csrc/gemv_kernels.h"
template __global__ void gemv_q4_qkv_rope_kernel<4, 2, 2>(QGemmArgs, const uint8_t *, RopeEpi, RmsSrc);
template __global__ void gemv_q4_gateup_silu_kernel<4, 2, 2>(QGemmArgs, const uint8_t *, int, float *, RmsSrc);
template __global__ void gemv_q4_rows_lds_kernel<4, 2, 0, 2>(QGemmArgs, const uint8_t *, int);
template __global__ void gemv_q4_rows_lds_kernel<4, 2, 1, 4>(QGemmArgs, const uint8_t *, int);
template __global__ void attn_gqa_split_kernel<128>(const float *, int, const uint32_t *, const float *, const float *, int, int, float, float *, const uint32_t *,
                                                    const uint32_t *, const uint32_t *, int64_t, QPre);
"""
STREAM_TU = """
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include "{root}/tts.cpp_amd/csrc/gemv_stream_kernels.h"
template __global__ void gemv_stream_kernel<4, PRO_F32, EPI_STORE>(GemmArgs, StreamMap);
template __global__ void gemv_stream_kernel<16, PRO_F32, EPI_STORE>(GemmArgs, StreamMap);
template __global__ void gemv_stream_kernel<4, PRO_F16, EPI_STORE>(GemmArgs, StreamMap);
template __global__ void gemv_stream_kernel<4, PRO_ATTN8, EPI_STORE>(GemmArgs, StreamMap);
template __global__ void gemv_stream_kernel<4, PRO_SILU, EPI_STORE>(GemmArgs, StreamMap);
template __global__ void gemv_stream_kernel<4, PRO_SILU, EPI_STORE, 4>(GemmArgs, StreamMap);
template __global__ void qgemv_stream_kernel<4, 2>(QGemmArgs, StreamMap);
template __global__ void qgemv_stream_kernel<8, 2>(QGemmArgs, StreamMap);
"""
DAC_TU = """
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <type_traits>
#include "{root}/tts.cpp_amd/csrc/dac_kernels.h"
#include "{root}/tts.cpp_amd/csrc/dac_b3_kernels.h"
template __global__ void conv_b3p_kernel<1, 4, 2, 2, 4, 1, 2, 2, SplitH2>(PConvArgs);   // round 6: the fp16 hi + lo planes are the default arithmetic
template __global__ void conv_b3p_kernel<1, 2, 4, 2, 4, 1, 2, 2, SplitH2>(PConvArgs);
template __global__ void convt_b3_kernel<8, 1, true, SplitH2>(ConvTArgs);
template __global__ void conv1d_mfma_kernel<7, 2, 2, 1, 4, 4>(ConvArgs);
"""
# dependent load groups a kernel may show in profiles/tools/isa_serial_loads.py (measured at the end of round 3, one of slack)
LIMITS = {
    r"gemm16_kernel<1, 1, 0, 1>": 4, r"gemm16_kernel<1, 1, 1, 1>": 4, r"gemm16_kernel<1, 3, 2, 1>": 3, r"gemm16_kernel<1, 2, 0, 1>": 3,
    r"gemm_tile_kernel<128, 128, 2, 4, 64, 4, 1>": 3, r"gemm_tile_kernel<128, 128, 2, 4, 64, 4, 2>": 3, r"gemm_tile_kernel<64, 64, 2, 4, 128, 3, 2>": 3,
    r"attn_short_kernel": 4, r"embed_rows_kernel": 4,
    r"conv_b3p_kernel<1, 4, 2, 2, 4, 1, 2, 2, SplitH2>": 17, r"conv_b3p_kernel<1, 2, 4, 2, 4, 1, 2, 2, SplitH2>": 10,
    # round 4: the dominant kernel and the restructured one-sequence kernels (Orpheus / Dia)
    r"attn_rows_kernel<8>": 4, r"attn_kernel(": 6, r"ln_rows_t_kernel<4, 0>": 2, r"ln_rows_t_kernelILi4ELi0E": 2,
    r"gemv_q4_qkv_rope_kernel<4, 2, 2>": 5, r"gemv_q4_gateup_silu_kernel<4, 2, 2>": 6, r"gemv_q4_rows_lds_kernel<4, 2, 0, 2>": 10, r"gemv_q4_rows_lds_kernel<4, 2, 1, 4>": 10, r"attn_gqa_split_kernel<128>": 3, r"attn_gqa_combine_kernel": 2,
    r"convt_b3_kernel<8, 1, true, SplitH2>": 3, r"conv1d_mfma_kernel<7, 2, 2, 1, 4, 4>": 14,   # 102 before its epilogue went to load / compute / store phases
}


def _compile(tmp_path, name, src):
    cu = tmp_path / f"{name}.hip"
    cu.write_text(src.format(root=ROOT))
    asm = tmp_path / f"{name}.s"
    p = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-Rpass-analysis=kernel-resource-usage",
                        "-o", str(asm), str(cu)], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    return asm, p.stderr


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_hot_kernels_have_no_scratch_and_no_serialized_load_chains(tmp_path):
    seen = {}
    for name, src in (("parler", PARLER_TU), ("dac", DAC_TU), ("llama", LLAMA_TU)):
        asm, remarks = _compile(tmp_path, name, src)
        scratch = [int(x) for x in re.findall(r"ScratchSize \[bytes/lane\]: (\d+)", remarks)]
        assert scratch and max(scratch) == 0, f"{name}: a kernel spills ({scratch})"
        out = subprocess.run([sys.executable, os.path.join(ROOT, "profiles", "tools", "isa_serial_loads.py"), str(asm), "1"],
                             capture_output=True, text=True, check=True).stdout
        filt = subprocess.run(["c++filt"], input=out, capture_output=True, text=True).stdout if shutil.which("c++filt") else out
        for line in filt.splitlines():
            m = re.match(r"\s*(\d+) dependent load groups\s+(.*)", line)
            if m:
                seen[m.group(2)] = int(m.group(1))
    checked = 0
    for pat, limit in LIMITS.items():
        hits = [(k, v) for k, v in seen.items() if pat in k]
        for k, v in hits:
            assert v <= limit, f"{k}: {v} dependent load groups (limit {limit}): a predicate crept back around a load?"
            checked += 1
    if shutil.which("c++filt"):
        assert checked >= 14, sorted(seen)   # (attn_gqa_split_kernel left the listing in round 5: its query, K and V requests are one group now)


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_streaming_gemv_requests_weights_and_rows_in_one_round_trip(tmp_path):
    """gemv_stream_kernel (every projection of a Dia step, the MLP of a Parler batch-1 step): the first weight chunk and the staging loads of the
    activation rows must be in flight together.  Until round 4 the weight loads sat under `if (t < tiles)`: the block's first result was copied on
    the way out, which put an `s_waitcnt vmcnt(7)` between the two groups — two dependent round trips at the head of ~110 launches per Dia step.
    The folding prologues (eight attention slices / silu * up) must also issue all of their loads before the first use, without scratch."""
    asm, remarks = _compile(tmp_path, "stream", STREAM_TU)
    scratch = [int(x) for x in re.findall(r"ScratchSize \[bytes/lane\]: (\d+)", remarks)]
    assert len(scratch) >= 7 and max(scratch) == 0, scratch
    # (round 5: the staging loads now go out BEFORE the weights and are waited for with a partial vmcnt on purpose — the order is checked by
    # test_one_sequence_kernels_request_their_staging_inputs_before_the_weights; what stays here is that nothing spills)


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_one_sequence_kernels_request_their_staging_inputs_before_the_weights(tmp_path):
    """Round 5: vmcnt retires in issue order.  Every kernel of the one-sequence chains (Parler batch 1: gemm16_kernel; Dia: gemv_stream_kernel; Orpheus:
    the Q4_0 GEMVs) requested its streamed weights first and its small staging inputs second — the wait for an input then retired the weights too and
    the prologue the weights were meant to fly under started one HBM round trip late (profiles/r05/isa_wait_order_before.txt lists 60 instances).
    profiles/tools/isa_wait_order.py must find none in the fp16 / Q4_0 instances the three chains launch."""
    tool = os.path.join(ROOT, "profiles", "tools", "isa_wait_order.py")
    par = PARLER_TU + "\n".join(f"template __global__ void gemm16_kernel<1, {pro}, {epi}, 1>(GemmArgs);" for pro, epi in
                               (("PRO_LN", "EPI_GELU"), ("PRO_F16", "EPI_RESID")))   # beside the four PARLER_TU holds: every launch of a batch-1 layer
    for name, src, must in (("parler_wo", par, "gemm16_kernel<1"), ("stream_wo", STREAM_TU, "gemv_stream_kernel"), ("llama_wo", LLAMA_TU, "gemv_q4")):
        asm, _ = _compile(tmp_path, name, src)
        assert must.split("<")[0] in open(asm).read()
        out = subprocess.run([sys.executable, tool, str(asm)], capture_output=True, text=True, check=True).stdout
        filt = subprocess.run(["c++filt"], input=out, capture_output=True, text=True).stdout if shutil.which("c++filt") else out
        # (qgemv_stream_kernel, the 5 .. 64-row integer form of round 6, is not a one-sequence kernel: its staging goes out before the weights too, but a
        #  slice larger than its straight-line staging is finished by loops behind the weight requests — by design, under the weights' flight)
        bad = [ln for ln in filt.splitlines() if must in ln and "gemm16_kernel<1, 6," not in ln and "qgemv_stream_kernel" not in ln]
        assert not bad, "\n".join(bad)
        if name == "parler_wo":
            # PRO_CROSS (the cross-attention in the out projection's prologue): the softmax over the first eight keys must start before the weights
            # land; the later chunks of a longer voice prompt (E > 8) are requested behind the weights and wait for them
            out = subprocess.run([sys.executable, tool, str(asm), "gemm16_kernelILi1ELi6E", "--first-use=v_exp_f32"], capture_output=True, text=True,
                                 check=True).stdout
            assert "gemm16_kernel" not in out, out


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_streaming_kernels_prefetch_overlaps_the_current_chunk(tmp_path):
    """Round 6 (profiles/r06/stream_loop_waits.txt): with the next chunk's loads under `if (next chunk exists)` the compiler's waits for the CURRENT chunk counted
    down to vmcnt(0) — the chunk just requested had to land before the current one was finished, no overlap.  The loops now issue unconditionally in the block
    in front of the MFMAs: every run of >= 8 weight loads that is followed by the MFMAs of a chunk must leave those MFMAs waiting with vmcnt >= 8 (eight loads =
    the newer chunk may still be in flight), in the fp16 and in the integer streaming kernel."""
    asm, remarks = _compile(tmp_path, "stream_ov", STREAM_TU)
    text = open(asm).read()
    checked = 0
    for sym in ("_Z18gemv_stream_kernelILi4ELi0ELi0ELi8EE", "_Z18gemv_stream_kernelILi16ELi0ELi0ELi8EE", "_Z19qgemv_stream_kernelILi4ELi2ELi1ELb0EE", "_Z19qgemv_stream_kernelILi8ELi2ELi1ELb0EE"):
        m = re.search(r"^" + sym + r"[^\n]*\n(.*?)^\.Lfunc_end", text, re.S | re.M)   # (a kernel has several s_endpgm: waves without a tile leave early)
        assert m, sym
        lines = m.group(1).splitlines()
        i, blocks = 0, 0
        while i < len(lines):
            if "global_load_dwordx4" in lines[i]:
                j, loads = i, 0
                while j < len(lines) and ("global_load_dwordx4" in lines[j] or not re.search(r"v_mfma|s_waitcnt vmcnt|s_cbranch|s_barrier|ds_write", lines[j])):
                    loads += "global_load_dwordx4" in lines[j]
                    j += 1
                if loads >= 8 and j < len(lines):   # what follows the run: MFMAs of a chunk (the loop) or LDS writes / a barrier (the staging prologue)
                    waits, mf, k = [], 0, j
                    while k < len(lines) and mf < 8 and not re.search(r"s_cbranch|s_barrier|ds_write|global_load", lines[k]):
                        w = re.search(r"s_waitcnt vmcnt\((\d+)\)", lines[k])
                        if w:
                            waits.append(int(w.group(1)))
                        mf += "v_mfma" in lines[k]
                        k += 1
                    if mf >= 4 and waits:
                        assert min(waits) >= 8, f"{sym}: MFMAs behind a prefetch wait with vmcnt({min(waits)}): the prefetched chunk is waited for"
                        blocks += 1
                i = max(j, i + 1)
            else:
                i += 1
        assert blocks >= 2, (sym, blocks)
        checked += blocks
    assert checked >= 8
