"""Host-side checks of the stream generator (project-nerf_amd/csrc/gen_stream_asm.py): the counted waits it emits
come from a model of the in-order LDS / vector-memory queues, and a modelling slip costs time silently (round 2:
every mask-word wait of the dgrad stream had degenerated to vmcnt(0))."""
import os
import re
import subprocess
import sys

import pytest

GEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "project-nerf_amd", "csrc", "gen_stream_asm.py")


def _generate(mode):
    env = {k: v for k, v in os.environ.items() if not k.startswith("GEN_")}
    env["GEN_MODES"] = mode
    r = subprocess.run([sys.executable, GEN], capture_output=True, text=True, env=env, cwd=os.path.dirname(GEN))
    assert r.returncode == 0, r.stderr
    return r.stdout


@pytest.mark.parametrize("mode,max_full_drains", [("bwd", 4), ("train", 1), ("bwd16", 4), ("train16", 1)])
def test_vector_memory_waits_are_counted(mode, max_full_drains):
    text = _generate(mode)
    waits = [int(n) for n in re.findall(r"s_waitcnt vmcnt\((\d+)\)", text)]
    assert len(waits) >= 15
    # vmcnt(0) only for operations issued before the pass (first mask words / first ring chunk)
    assert sum(1 for n in waits if n == 0) <= max_full_drains, waits
    assert max(waits) <= 63


def test_bf16_image_streams_store_two_halves_per_tile():
    """bf16 training images (the default): every stashed m-tile leaves as two 16-byte stores per lane (2-KiB blocks),
    straight from the operand registers -- no 8-bit conversion anywhere in the stream."""
    for mode, tiles in (("train16", 8 * 8 + 8 + 4), ("bwd16", 4 + 8 + 8 * 8)):
        text = _generate(mode)
        assert "v_cvt_scalef32" not in text and "HW_REG_MODE" not in text
        assert len(re.findall(r"global_store_dwordx4 v89, v\[\d+:\d+\], s\[90:91\] nt", text)) == tiles
        assert len(re.findall(r"global_store_dwordx4 v89, v\[\d+:\d+\], s\[90:91\] offset:128 nt", text)) == tiles
        assert "s_lshl_b64 s[96:97], s[96:97], 9" in text


@pytest.mark.parametrize("mode", ["bwd", "bwd16"])
def test_dgrad_waits_once_per_masked_tile(mode):
    text = _generate(mode)
    # 68 of the 76 tiles carry a ReLU mask word (B_VIEW's 8 do not); the first three words are loaded by the previous pass
    n_mask_loads = len(re.findall(r"global_load_ushort v9[1-4], v90, s\[92:93\]", text))
    assert n_mask_loads >= 68, n_mask_loads
    assert "GEN_CONFIG D=4 NO=\n" in text            # default configuration: no timing ablation leaked into the build


def test_inference_stream_with_64_samples_per_wave():
    """mode infer64: the same fragment stream as infer16 (1184 fragments), every fragment feeding FOUR MFMAs on distinct
    accumulators; B operands of the hidden layers come from AGPRs, nat operands from the x / d inputs; an accumulator's first MFMA
    takes the group's bias registers as C; every epilogue pair reaches its AGPR through v_accvgpr_write_b32; two bias reads per
    group; DMA pieces of four waves (4 KiB apart)."""
    t16, t64 = _generate("infer16"), _generate("infer64")
    mfma16 = re.findall(r"v_mfma_f32_16x16x32_bf16 (\S+), %\[w(\d)\], (\S+), (\S+)\\n", t16)
    mfma64 = re.findall(r"v_mfma_f32_16x16x32_bf16 (\S+), %\[w(\d)\], (\S+), (\S+)\\n", t64)
    assert len(mfma64) == 2 * len(mfma16) == 4 * 1184
    reads16 = re.findall(r"ds_read_b128 %\[w\d\], %\[ab[01]\] offset:(\d+)", t16)
    reads64 = re.findall(r"ds_read_b128 %\[w\d\], %\[ab[01]\] offset:(\d+)", t64)
    assert reads16 == reads64 and len(reads64) == 1184                  # the same fragments in the same order, once per 64 samples
    for i in range(0, len(mfma64), 4):                                  # four MFMAs per fragment: one window register, four accumulators
        quad = mfma64[i:i + 4]
        assert len({m[1] for m in quad}) == 1 and len({m[0] for m in quad}) == 4
        assert all(m[2].startswith("a[") or m[2].startswith("%[x") or m[2].startswith("%[d") for m in quad)
        assert all(m[3] == m[0] or re.fullmatch(r"v\[1(6[0-9]|7[0-5]):1(6[0-9]|7[0-5])\],?", m[3]) for m in quad)   # C: itself or v[160:175]
    assert len(re.findall(r"v_accvgpr_write_b32 a\d+, v(89|9[0-5])", t64)) == len(re.findall(r"v_cvt_pk_bf16_f32 v(89|9[0-5]),", t64)) == 16 * 76
    assert len(re.findall(r"ds_read_b128 v\[1(6[0-9]|7[0-5]):1(6[0-9]|7[0-5])\], %\[bb\]", t64)) == 2 * 78
    dma = [int(x, 16) for x in re.findall(r"v_add_u32 v88, (0x[0-9a-f]+), %\[voff\]", t64)]
    steps = [b - a for a, b in zip(dma, dma[1:])]
    assert dma and sum(1 for d in steps if d == 4096) > 0.9 * len(steps) and 8192 not in steps      # four waves x 1 KiB per piece
    assert "GEN_CONFIG D=4 NO=\n" in t64
