"""A short leg of every seeded campaign under tools/fuzz_*.py with the GPU suite (the campaigns themselves run hundreds
of cases on request): the HIP path against the CPU oracle on random box sets, scene seeds and tree rules, render
parameters, degenerate rays and degenerate scenes, and trees whose boxes do not bound their contents. A campaign prints
what differs and returns non-zero."""
import importlib
import os
import sys

import pytest

pytestmark = pytest.mark.gpu

TOOLS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tools")

CAMPAIGNS = [("fuzz_bvh_build", 60), ("fuzz_config_scenes", 2), ("fuzz_render_params", 60), ("fuzz_evil_rays", 10),
             ("fuzz_degenerate_scenes", 40), ("fuzz_per_axis", 5)]


@pytest.mark.parametrize("name,cases", CAMPAIGNS, ids=[c[0] for c in CAMPAIGNS])
def test_campaign(gpu, name, cases):
    if TOOLS not in sys.path:
        sys.path.insert(0, TOOLS)
    mod = importlib.import_module(name)
    assert mod.main(cases) == 0, name
