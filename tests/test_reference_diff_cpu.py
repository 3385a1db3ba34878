"""Live differential check of the HOST LOGIC against the real reference, where the reference is present (the build container:
/root/reference); skipped everywhere else -- the committed goldens of tests/golden/ are the portable form of the same evidence.
Runs a few seeded rounds of tools/ref_diff_fuzz.py: both packages in one process on the CPU, this engine on the stand-ins of
tests/cpu_engine.py, random stage configurations, every stored score and factor tensor compared."""
import importlib.util
import os
import random
import sys
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REFERENCE = "/root/reference/kronfluence"

pytestmark = pytest.mark.skipif(not os.path.isdir(REFERENCE), reason="the reference is only present in the build container")


@pytest.fixture(scope="module")
def fuzz():
    before = list(sys.path)
    spec = importlib.util.spec_from_file_location("ref_diff_fuzz", os.path.join(ROOT, "tools", "ref_diff_fuzz.py"))
    module = importlib.util.module_from_spec(spec)
    try:
        spec.loader.exec_module(module)
    except ImportError as exc:   # a dependency of the reference that the shims do not cover
        pytest.skip(f"the reference does not import here: {exc}")
    yield module
    sys.path[:] = before


@pytest.mark.parametrize("seed", [3, 4])
def test_random_stage_configurations_match_the_reference(fuzz, seed, cpu_engine):
    from kronfluence_amd.utils.state import State

    rng = random.Random(seed)
    for _ in range(5):
        cfg = fuzz.draw(rng)
        with tempfile.TemporaryDirectory() as a, tempfile.TemporaryDirectory() as b:
            want = fuzz.run(fuzz.ref_pkg, cfg, a, ours=False)
            State._reset_state()
            got = fuzz.run(fuzz.our_pkg, cfg, b, ours=True)
        assert set(want) == set(got), cfg
        for key, tensor in want.items():
            assert tensor.shape == got[key].shape, (key, cfg)
            error = float((got[key] - tensor).norm() / tensor.norm().clamp_min(1e-300))
            assert error <= 5e-5, (key, error, cfg)   # the stand-ins store fp32, like the engine
