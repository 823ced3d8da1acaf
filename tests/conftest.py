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
    if PARITY_REPORT:
        terminalreporter.write_sep("-", "parity report")
        for line in PARITY_REPORT:
            terminalreporter.write_line(line)
