"""A bounded run of the randomised GPU-vs-oracle sweep (tests/fuzz_parity.py) under `pytest -m gpu`: scenes, instance
sets, materials, debug options, depth limits, image sizes, both pipelines, both cube filters -- every draw bit-exact
in image and ray counts."""
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed", [11, 12])
def test_bounded_fuzz_parity(gpu, oracle, capi, seed):
    import fuzz_parity
    assert fuzz_parity.run(100, seed, gpu, verbose=False) is None
