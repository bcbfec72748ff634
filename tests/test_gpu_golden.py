"""The reference's own integration tests, run on the HIP path through the C ABI (`-m gpu`)."""
import pytest

import golden_cases as G

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def engine():
    import skx_engine as E
    E.load_library()
    E.default_context()          # raises when there is no usable device: there is no fallback
    return E


@pytest.mark.parametrize("case", G.ALL_CASES, ids=lambda c: c.__name__)
def test_reference_case(engine, case, tmp_path):
    case(engine, tmp_path)
