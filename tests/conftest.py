import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` through gpurun)")


def pytest_collection_modifyitems(config, items):
    import torch

    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def golden():
    import torch

    def load(name):
        return torch.load(os.path.join(GOLDEN, name), map_location="cpu", weights_only=False)

    return load


def rel_l2(a, b):
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


# ---------------------------------------------------------------------------------------------- tolerance contract
# bf16 storage => a tensor is compared with the fp32 reference result next to the yard-stick err_ref = error of
# the reference itself run in bf16 (tests/golden/err_ref.pt, *_bf16 entries of the head fixtures, or the oracle evaluated in
# bf16 inside the test); fp32 scalars (losses) are held to 1e-3 relative plus the same yard-stick allowance.
_PARITY_LOG = []


def bound(err_ref):
    return 1.5 * err_ref + 2e-3


def check_tensor(name, ours, ref, err_ref):
    """assert rel-L2(ours, ref) <= 1.5 * err_ref + 2e-3 and log the measured pair (gpurun_out/parity_report.json)."""
    e = rel_l2(ours, ref)
    _PARITY_LOG.append(dict(name=name, err=e, err_ref=err_ref, bound=bound(err_ref)))
    assert e <= bound(err_ref), f"{name}: rel-L2 {e:.3e} > 1.5 * err_ref({err_ref:.3e}) + 2e-3"
    return e


def check_scalar(name, ours, ref, ref_bf16_abs_err=0.0, rtol=1e-3):
    """fp32 scalar (a loss): |ours - ref| <= 1.5 * |reference_in_bf16 - ref| + 1e-3 * |ref|."""
    ours, ref = float(ours), float(ref)
    lim = 1.5 * float(ref_bf16_abs_err) + rtol * abs(ref)
    _PARITY_LOG.append(dict(name=name, err=abs(ours - ref) / max(abs(ref), 1e-30), err_ref=float(ref_bf16_abs_err) / max(abs(ref), 1e-30),
                            bound=lim / max(abs(ref), 1e-30), scalar=True))
    assert abs(ours - ref) <= lim, f"{name}: |{ours:.6f} - {ref:.6f}| = {abs(ours - ref):.3e} > {lim:.3e}"


def pytest_sessionfinish(session, exitstatus):
    if not _PARITY_LOG:
        return
    import json
    out = os.path.join(ROOT, "gpurun_out")
    try:
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, "parity_report.json"), "w") as f:
            json.dump(_PARITY_LOG, f, indent=1)
    except OSError:
        pass
