import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "reference: needs /root/reference (build container only)")


# Lines the parity tests want in the log even when they pass (mismatch counts, measured errors): printed once at the end.
PARITY_REPORT = []


def pytest_terminal_summary(terminalreporter):
    mod = sys.modules.get("tests.test_gpu_parity") or sys.modules.get("test_gpu_parity")
    worst = getattr(mod, "GRAD_WORST", None)
    if worst and worst["rel"] > 0.0:
        PARITY_REPORT.append("gradients: worst deviation from autograd over the oracle %.2e of the gradient's largest entry (%s); "
                             "asserted 2e-4 and, regression level, 2e-5" % (worst["rel"], worst["name"]))
    if PARITY_REPORT:
        terminalreporter.write_sep("-", "parity report")
        for line in PARITY_REPORT:
            terminalreporter.write_line(line)
