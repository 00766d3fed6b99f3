"""The opt-in kernel variants stay parity-green: each is selected by an environment variable the library latches at its first use, so
the parity tests of the default kernels are re-run in a child process with the variable set.

  * K9e (`spconv_fwd_wide_kernel`: 128-row blocks, accumulators kept in the unit of the row being multiplied) —
    FSF_PLANES_WIDE_MIN_ROWS=1;
  * K9g (the same kernel template on 96-row blocks, half the cells' fragments in registers at a time: three workgroups per CU) —
    FSF_PLANES_R96_MIN_ROWS=1;
  * K9f (`spconv_fwd_tri_kernel`: 192-row workgroups of twelve waves sharing the weight fragments through LDS) —
    FSF_PLANES_TRI_MIN_ROWS=1;
  * K9c as the only plane kernel (K9d off) — FSF_PLANES_PIPE=0;
  * K22b (`linear_norm_act_f16_kernel`: f16 planes x3, line-coalesced x through a wave-private LDS tile) — FSF_K22_F16=1;
  * 12-wave K22 workgroups — FSF_K22_WIDE_MIN_ROWS=1.
DESIGN.md section 5 has what each measured (none is the default: slower or neutral)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_child(env_extra, select):
    env = dict(os.environ, **env_extra)
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_hip_ops.py"), "-m", "gpu", "-x", "-q", "-k", select,
                        "-p", "no:cacheprovider"], capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    tail = (r.stdout + r.stderr)[-2500:]
    assert r.returncode == 0, tail
    assert " passed" in r.stdout and "failed" not in r.stdout, tail


@pytest.mark.parametrize("env", [dict(FSF_PLANES_WIDE_MIN_ROWS="1"), dict(FSF_PLANES_PIPE="0"), dict(FSF_PLANES_TRI_MIN_ROWS="1"),
                                 dict(FSF_PLANES_R96_MIN_ROWS="1")], ids=["K9e", "K9c", "K9f", "K9g"])
def test_plane_kernel_variants(device, env):
    run_child(env, "spconv_forward_planes")


@pytest.mark.parametrize("env", [dict(FSF_K22_F16="1"), dict(FSF_K22_WIDE_MIN_ROWS="1")], ids=["K22b", "K22-12wave"])
def test_linear_norm_act_variants(device, env):
    run_child(env, "linear_norm_act")
