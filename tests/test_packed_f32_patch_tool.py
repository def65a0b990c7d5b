"""CPU: tools/probes/asm_patch_build.py (docs/PACKED_F32_HAZARD.md, round 5) still finds the loop it patches.

The experiment that pins the packed-f32 fault on `v_pk_fma_f32 ... op_sel:[0,1,0]` edits the compiler's assembly of ONE loop of
k_head_bwd; a compiler or source change that moves that loop would silently turn the recipe into a no-op.  This test compiles
sac.hip to gfx950 assembly with the vectorized flags (as the tool does), applies the patches and checks what they changed."""
import importlib.util
import os
import subprocess
import sys
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rl-x_amd"))
import build as rlx_build  # noqa: E402


@pytest.fixture(scope="module")
def tool_and_asm():
    if not os.path.exists(rlx_build.HIPCC):
        pytest.skip("hipcc not installed")
    spec = importlib.util.spec_from_file_location("asm_patch_build", os.path.join(ROOT, "tools", "probes", "asm_patch_build.py"))
    tool = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(tool)
    vec = [f for f in rlx_build.CFLAGS if f not in rlx_build._NOVEC]
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "sac.s")
        r = subprocess.run([rlx_build.HIPCC] + vec + ["--cuda-device-only", "-S", os.path.join(rlx_build.CSRC, "sac.hip"), "-o", out],
                           capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        return tool, open(out).read()


def test_the_failing_loop_is_where_the_tool_looks_for_it(tool_and_asm):
    tool, asm = tool_and_asm
    _, body = tool.patch_text(asm, "none")
    assert body.count("v_pk_fma_f32") == 16 and body.count("ds_read_b128") == 8
    assert body.count("op_sel:[0,1,0]") == 8 and body.count("op_sel_hi:[1,0,1]") == 8      # column 2 / column 1 of the two-column loop


def test_nosel_replaces_exactly_the_routed_operand(tool_and_asm):
    tool, asm = tool_and_asm
    patched, body = tool.patch_text(asm, "nosel")
    assert "op_sel:[0,1,0]" not in body and body.count("v[88:89]") == 8 and body.count("v_mov_b32_e32 v8") == 2
    assert body.count("op_sel_hi:[1,0,1]") == 8                                             # column 1 untouched
    assert ".amdhsa_next_free_vgpr 90" in patched
    _, keep = tool.patch_text(asm, "selcopy")
    assert keep.count("op_sel:[0,1,0]") == 8 and keep.count("v[88:89]") == 8                # the control keeps the modifier


def test_wait_state_patches_only_insert(tool_and_asm):
    tool, asm = tool_and_asm
    base = tool.patch_text(asm, "none")[1].split("\n")
    for patch, extra in (("nop15_both", 2), ("nop15_mid", 1), ("wait_first", 2), ("vnop_both", 8)):
        body = tool.patch_text(asm, patch)[1].split("\n")
        assert len(body) == len(base) + extra, patch
        assert [l for l in body if l in base] == base or len([l for l in body if l not in base]) <= extra, patch
