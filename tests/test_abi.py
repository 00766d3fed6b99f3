"""CPU: the C-ABI library builds for gfx950, loads, and exports every symbol include/fsf_hip.h declares.
No compute calls (there is no GPU in the build container)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "fsf_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(fsf_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_hot_path():
    syms = declared_symbols()
    for must in ["fsf_voxelize_dynamic", "fsf_voxelize_divfloor", "fsf_unique_rows", "fsf_segment_reduce",
                 "fsf_segment_reduce_backward", "fsf_gather_rows", "fsf_voxel2point", "fsf_project_gather_mask",
                 "fsf_cam_select_score", "fsf_rulebook_subm", "fsf_rulebook_strided", "fsf_rulebook_to_pairs",
                 "fsf_spconv_forward", "fsf_connected_components", "fsf_ingroup_rank"]:
        assert must in syms


def test_library_builds_and_exports_every_declared_symbol():
    from fullysparsefusion_amd import build

    lib_path = build.build()
    assert os.path.exists(lib_path)
    lib = ctypes.CDLL(lib_path)
    missing = [s for s in declared_symbols() if not hasattr(lib, s)]
    assert not missing, f"declared in include/fsf_hip.h but not exported: {missing}"
    lib.fsf_status_string.restype = ctypes.c_char_p
    assert lib.fsf_status_string(0) == b"ok"
    header = int(re.search(r"#define\s+FSF_ABI_VERSION\s+(\d+)", open(os.path.join(ROOT, "include", "fsf_hip.h")).read()).group(1))
    assert lib.fsf_abi_version() == header  # (the loader in fullysparsefusion_amd/_lib.py refuses a library of another version)


def test_wrapper_argtypes_cover_the_header():
    from fullysparsefusion_amd import hip_ops

    declared = set(declared_symbols()) - {"fsf_status_string", "fsf_abi_version"}
    assert declared == set(hip_ops._ARGTYPES), declared ^ set(hip_ops._ARGTYPES)


def test_product_path_fails_loudly_without_a_gpu():
    import torch

    from fullysparsefusion_amd import hip_ops
    from fullysparsefusion_amd._lib import FsfHipError

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(FsfHipError):
        hip_ops.unique_rows(torch.zeros((4, 4), dtype=torch.int64))
    with pytest.raises(FsfHipError):
        hip_ops.voxelize_dynamic(torch.zeros((4, 5)), (0.2, 0.2, 0.2), [-51.2, -51.2, -5, 51.2, 51.2, 3], [512, 512, 40])


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "fullysparsefusion_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), os.path.join(dirpath, f)


def test_prepared_linear_weights_carry_their_format_as_a_tag_not_as_a_size():
    """ADVICE r5 / VERDICT r5 next-7: hip_ops.linear_norm_act picks the entry point (bf16 x 6 or K22f's f16 x 3) from the tag the
    preparing call attached; nothing in the package infers it from the buffer's size any more, and an untagged buffer is refused."""
    import torch

    from fullysparsefusion_amd import hip_ops

    src = open(os.path.join(ROOT, "fullysparsefusion_amd", "hip_ops.py")).read()
    assert "& 1023" not in src
    with pytest.raises(hip_ops.FsfHipError, match="format tag"):
        hip_ops.linear_weight_is_f16(torch.zeros(1280, dtype=torch.uint8))
    t = torch.zeros(1280, dtype=torch.uint8)
    setattr(t, hip_ops._FMT_ATTR, "f16x3")
    assert hip_ops.linear_weight_is_f16(t)
    csrc = open(os.path.join(ROOT, "fullysparsefusion_amd", "csrc", "linear_norm_act.hip")).read()
    assert "LNA_F16_TAG" in csrc and "__builtin_trap()" in csrc  # the f16 kernels verify the header's tag word


def test_every_entry_point_of_the_header_is_named_in_integration_md():
    """INTEGRATION.md shows the reference-side binding of every entry point (size functions are covered by one line)."""
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    missing = [s for s in declared_symbols() if not s.endswith("_bytes") and s not in doc]
    assert missing == [], missing
