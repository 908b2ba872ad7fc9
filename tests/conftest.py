import os
import sys

import pytest

# the tests select kernel variants through the library's measurement hooks (WG_FLOW_BLOCK, WG_FLOW_ENV, WG_SUMS ...), which
# the library honours only with this switch (wg_api.hip: wg_hook)
os.environ["WG_DEBUG_HOOKS"] = "1"

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle_lib():
    from oracle import oracle as om
    om.build()
    return om
