"""pytest configuration: registers the `gpu` marker and puts the repo root and the
test-only oracle on the import path."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # an A/B build left in the environment would make every test (and every measured parity bound) run against another
    # binary than the committed sources: the suite refuses to start
    if os.environ.get("EKF_LIB_PATH"):
        raise pytest.UsageError("EKF_LIB_PATH is set (" + os.environ["EKF_LIB_PATH"] + "): the test suite only runs against the in-tree "
                                "library built from the committed sources; unset it")


def pytest_sessionfinish(session, exitstatus):
    """EKF_PARITY_CEILING_ONLY switches every measured 10 x bound of tests/helpers.bound() off (a measurement run in front
    of tools/update_parity_bounds.py).  A test session that ran with it leaking in from the environment must not read as
    a green parity run: unless the run also logs its measurements (EKF_PARITY_LOG: the measurement run of the
    tool), the session FAILS."""
    if os.environ.get("EKF_PARITY_CEILING_ONLY"):
        msg = ("EKF_PARITY_CEILING_ONLY is set: every parity site ran against its loose hand-written ceiling only, not against "
               "its measured bound (tests/golden/parity_bounds.json)")
        tr = session.config.pluginmanager.get_plugin("terminalreporter")
        if tr is not None:
            tr.write_line("WARNING: " + msg, red=True, bold=True)
        if not os.environ.get("EKF_PARITY_LOG") and exitstatus == 0:
            session.exitstatus = 1
