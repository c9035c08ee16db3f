import os
import sys
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    import oracle_api
    return oracle_api.load()


def pytest_terminal_summary(terminalreporter):
    """which end-to-end comparisons were exact and which used the eps-sliver rule (tests/poly_harness.py)"""
    try:
        import poly_harness as ph
    except Exception:
        return
    if not ph.PARITY_MODES:
        return
    sl = ["%s=%s" % (t, m) for t, m in ph.PARITY_MODES if m != "exact"]
    terminalreporter.write_line("parity modes: %d comparisons exact, %d sliver%s" % (len(ph.PARITY_MODES) - len(sl), len(sl), (": " + ", ".join(sl)) if sl else ""))
    out = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(out):
        import json
        with open(os.path.join(out, "parity_modes.json"), "w") as f:
            json.dump([{"test": t, "mode": m} for t, m in ph.PARITY_MODES], f, indent=0)
