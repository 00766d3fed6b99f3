import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def golden_cases(npz):
    """Group 'case__field' keys of a golden file into {case: {field: array}}."""
    cases = {}
    for k in npz.files:
        case, field = k.split("__", 1)
        cases.setdefault(case, {})[field] = npz[k]
    return cases


@pytest.fixture(scope="session")
def device():
    import torch

    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    return torch.device("cuda:0")


def build_test_fsf():
    """The FSF detector every module-level parity test uses (CPU, eval mode): reference config, seed 0, the zero-initialised
    image-branch Linear (FSF.py:142-143) perturbed so that the fusion is exercised, BN running statistics away from (0, 1)
    so that the fused conv epilogue is really tested.  Deterministic on the CPU generator — the full-size golden
    (tests/golden/make_fullsize_golden.py) is generated from the same function and stores a parameter checksum."""
    import torch

    from fullysparsefusion_amd import mmdet3d_plugin
    from fullysparsefusion_amd.compat import Config

    torch.manual_seed(0)
    cfg = Config.fromfile(os.path.join(ROOT, "configs", "fsf_nuscenes.py"))
    model = mmdet3d_plugin.build_model(cfg.model).eval()
    torch.nn.init.normal_(model.segmentor_updated_mlp[-1].weight, std=0.05)
    for m in model.modules():
        if isinstance(m, torch.nn.BatchNorm1d):
            m.running_mean.normal_(0, 0.1)
            m.running_var.uniform_(0.5, 1.5)
            m.weight.data.uniform_(0.8, 1.2)
            m.bias.data.normal_(0, 0.1)
    return model


def build_av2_fsf(perturb_image_branch=False):
    """The Argoverse-2 detector (configs/fsf_av2.py = the model part of the reference's FSF_AV2_config.py) behind
    tests/golden/av2_segmentor_150k.npz: fixed seed, BN running statistics away from (0, 1).  `perturb_image_branch`
    un-zeroes the image branch's last Linear (FSF.py:142-143) from its own generator — the segmentor, which the golden's
    parameter checksum covers, is untouched."""
    import torch

    from fullysparsefusion_amd import mmdet3d_plugin
    from fullysparsefusion_amd.compat import Config

    torch.manual_seed(11)
    cfg = Config.fromfile(os.path.join(ROOT, "configs", "fsf_av2.py"))
    model = mmdet3d_plugin.build_model(cfg.model).eval()
    for m in model.modules():
        if isinstance(m, torch.nn.BatchNorm1d):
            m.running_mean.normal_(0, 0.1)
            m.running_var.uniform_(0.5, 1.5)
    if perturb_image_branch:
        g = torch.Generator().manual_seed(5)
        w = model.segmentor_updated_mlp[-1].weight
        w.data.copy_(torch.randn(w.shape, generator=g) * 0.05)
    return model


def param_checksum(model):
    """float64 sum of |p| over parameters and buffers: detects a model that differs from the one a golden was made with."""
    import torch

    tot = 0.0
    for t in list(model.parameters()) + list(model.buffers()):
        if t.is_floating_point():
            tot += float(t.detach().double().abs().sum())
    return tot
