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
