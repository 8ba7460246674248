import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "gan-pytorch_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


@pytest.fixture(scope="session")
def gpu():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("test marked gpu but no GPU is visible")
    return torch.device("cuda:0")


def pytest_terminal_summary(terminalreporter):
    """Free-running parameter comparisons allow a small share of elements beyond 1e-4 (golden_util.compare_network):
    say how much of that allowance the run actually used, so a regression inside the budget is visible."""
    try:
        import golden_util as gu
    except ImportError:
        return
    rows = gu.OUTLIER_REPORT
    if not rows:
        return
    used = [(n / float(size), label, key, n, size) for label, key, n, size in rows if n]
    terminalreporter.write_line("parameter tensors compared free-running: %d, with elements beyond tolerance: %d "
                                "(allowance: %.2f%% of a tensor, at least one element)" %
                                (len(rows), len(used), 100 * gu.OUTLIER_FRACTION))
    terminalreporter.write_line("   largest single-element difference: %.2e (%s)" % tuple(gu.LARGEST_DRIFT))
    for share, label, key, n, size in sorted(used, reverse=True)[:8]:
        terminalreporter.write_line("   %-12s %-28s %d of %d (%.2f%%)" % (label, key, n, size, 100 * share))
