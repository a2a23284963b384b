"""The generated gfx950 code of the matrix scan against two rules about the operands of matrix instructions
(tools/mfma_operand_lint.py): nothing writes the A/B registers of a v_mfma between its issue and the first read of
its accumulators, and no vector instruction writes a source register directly ahead of the v_mfma that reads it.
Green on the library's three builds; red on the rejected form of the tile loop that loads the next tile's fragments
into the registers the matrix instructions in flight were given."""
import os
import shutil
import subprocess
import sys

import pytest

from conftest import ROOT

sys.path.insert(0, os.path.join(ROOT, "tools"))
import mfma_operand_lint as lint  # noqa: E402

CSRC = os.path.join(ROOT, "raytracer_amd", "csrc")
pytestmark = pytest.mark.skipif(shutil.which("hipcc") is None, reason="hipcc not on PATH")


def _isa(tmp_path, name, defs=""):
    d = str(tmp_path / name)
    subprocess.check_call(["make", "-s", "-C", CSRC, "isa", "ISADIR=" + d, "ISADEF=" + defs],
                          stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return [os.path.join(d, "rt_trace_%s.s" % b) for b in ("exact", "strict", "fast")]


def test_library_builds_keep_matrix_operands_intact(tmp_path):
    for path in _isa(tmp_path, "kept"):
        assert lint.count_mfma(path) == 32            # 2 modes x 2 matrix variants x 2 unrolled tiles x 4
        bad = lint.lint(path)
        assert not bad, ["%s: line %d %s | line %d %s" % (os.path.basename(path), m.line_no, m.text, i.line_no, i.text)
                         for _, m, i in bad[:5]]


def test_rejected_tile_loop_is_flagged(tmp_path):
    exact = _isa(tmp_path, "pipelined", "-DRT_TILE_LOOP_PIPELINED -DRT_NO_OPERAND_FENCE")[0]
    bad = lint.lint(exact)
    assert bad and all(i.mnemonic.startswith(("ds_read", "global_load")) for _, _, i in bad)     # the next tile's fragment reads
    # the same loop with the operand fence compiles to code that keeps the rule
    fenced = _isa(tmp_path, "pipelined_fenced", "-DRT_TILE_LOOP_PIPELINED")[0]
    assert not lint.lint(fenced)


def test_lint_sees_an_adjacent_write(tmp_path):
    src = tmp_path / "k.s"
    src.write_text("k:\n\tv_mov_b32_e32 v7, v1\n\ts_waitcnt lgkmcnt(0)\n\tv_mfma_f32_32x32x16_f16 v[16:31], v[4:7], v[8:11], 0\n"
                   "\ts_nop 10\n\tv_max_f32_e32 v0, v16, v17\n\ts_endpgm\n")
    bad = lint.lint(str(src))
    assert len(bad) == 1 and bad[0][2].text.startswith("v_mov_b32")
    src.write_text("k:\n\tv_mov_b32_e32 v7, v1\n\tv_nop\n\tv_mfma_f32_32x32x16_f16 v[16:31], v[4:7], v[8:11], 0\n"
                   "\ts_nop 10\n\tv_max_f32_e32 v0, v16, v17\n\tds_read_b128 v[4:7], v3\n\ts_endpgm\n")
    assert not lint.lint(str(src))
