"""Host-side plumbing that runs without a GPU: the one environment hook of the host side (SGG_OPTIONS -> lib.options_from_env) and the
kernel-timeline analysis the round's schedule work rests on (scripts/trace_timeline.py, plain and --gated windows)."""
import os
import subprocess
import sys

import pytest

import sgg_amd  # noqa: F401
from sgg_amd.lib import DEFAULT_OPTIONS, SggError, options_from_env

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_options_from_env_types(monkeypatch):
    monkeypatch.setenv("SGG_OPTIONS", "presplit=0, g_early_cus=24,ln_fusion_skip=4+5,halo_pc=off")
    o = options_from_env()
    assert o["presplit"] is False and o["halo_pc"] is False          # booleans: 0 / false / no / off
    assert o["g_early_cus"] == 24 and isinstance(o["g_early_cus"], int)
    assert o["ln_fusion_skip"] == (4, 5)                             # tuples of conv indices: '+'-separated
    untouched = {k: v for k, v in o.items() if k not in ("presplit", "halo_pc", "g_early_cus", "ln_fusion_skip")}
    assert untouched == {k: v for k, v in DEFAULT_OPTIONS.items() if k in untouched}
    monkeypatch.setenv("SGG_OPTIONS", "")
    assert options_from_env() == dict(DEFAULT_OPTIONS)


def test_options_from_env_rejects_unknown_names(monkeypatch):
    monkeypatch.setenv("SGG_OPTIONS", "no_such_option=1")
    with pytest.raises(SggError):
        options_from_env()


def _trace(path, rows):
    with open(path, "w") as f:
        f.write('"Kind","Agent_Id","Queue_Id","Kernel_Name","Start_Timestamp","End_Timestamp","Grid_Size_X"\n')
        for q, name, a, b in rows:
            f.write('"KERNEL_DISPATCH",1,%d,"%s",%d,%d,256\n' % (q, name, a, b))


def _run(args):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "trace_timeline.py")] + args, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return r.stdout


def test_trace_timeline_gated_window(tmp_path):
    """--gated: the window is the LAST step, from the end of its spin kernel to its last kernel; a matrix kernel = a conv_* symbol."""
    us = 1000
    rows = [(1, "void at::native::spin_kernel(long)", 0, 100 * us),
            (1, "void conv_halo3_pc_kernel<true, false, true, 2>(HaloParams)", 100 * us, 300 * us),
            (1, "adam_kernel(float*)", 300 * us, 320 * us),
            (1, "void at::native::spin_kernel(long)", 400 * us, 500 * us),
            (1, "void conv_s2_kernel<false, true, 7, false, false, true, 8>(S2Params)", 500 * us, 900 * us),
            (2, "void ln_apply_elu_s16_kernel<false>(float const*)", 800 * us, 1000 * us),      # 100 us beside the conv, 100 us alone
            (1, "void conv_halo3_pc_kernel<true, false, true, 2>(HaloParams)", 1100 * us, 1500 * us),      # 100 us with nothing resident
            (1, "adam_kernel(float*)", 1500 * us, 1550 * us)]
    p = str(tmp_path / "t.csv")
    _trace(p, rows)
    out = _run([p, "--gated", "--gantt"])
    assert "window 1.05 ms = 1 steps" in out, out
    assert "a matrix (conv) kernel resident :   0.80 ms" in out, out
    assert "only other kernels resident     :   0.15 ms" in out, out
    assert "nothing resident                :   0.10 ms" in out, out
    assert "q2 ln_apply_elu_s16_kernel<false>" in out                # the Gantt lines carry the queue


def test_trace_timeline_plain_window(tmp_path):
    """Without --gated the steps are delimited by the Adam launches (two per G+D step)."""
    us = 1000
    rows = []
    t = 0
    for step in range(3):
        for upd in range(2):
            rows.append((1, "void conv_s2_kernel<true, true, 7, false, false, true, 4>(S2Params)", t, t + 400 * us))
            rows.append((1, "adam_kernel(float*)", t + 400 * us, t + 500 * us))
            t += 500 * us
    p = str(tmp_path / "t.csv")
    _trace(p, rows)
    out = _run([p, "2"])
    assert "= 2 steps of 1.00 ms" in out, out
    assert "a matrix (conv) kernel resident :   0.80 ms per step" in out, out
