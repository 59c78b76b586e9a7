"""Pin the oracle (oracle/spliser_oracle.c + the host site-table code) to outputs of the real
SpliSER v0.1.8, committed under tests/golden/ by tests/golden/make_golden.py."""
import pytest

from conftest import golden_cases
import helpers


@pytest.mark.parametrize("case,variant,opts", golden_cases(), ids=lambda v: v if isinstance(v, str) else "")
def test_oracle_reproduces_reference_tsv(case, variant, opts, oracle_lib):
    text, rows = helpers.run_case(case, opts, helpers.oracle_engine(oracle_lib))
    ref_text, ref_rows = helpers.expected(case, variant)
    assert text == ref_text
    helpers.assert_rows_match(rows, ref_rows, bool(opts.get("cryptic")))
