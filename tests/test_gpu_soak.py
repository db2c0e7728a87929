"""1500 mixed calls on one long-lived handle and on short-lived ones: results stay finite, free device memory settles."""
import os
import runpy

import pytest

pytestmark = pytest.mark.gpu


def test_soak():
    runpy.run_path(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "soak.py"), run_name="__main__")
