"""Randomised parity sweep (tests/fuzz.py) as a GPU test: 150 random problems (the campaigns of hundreds run as evidence: tools/fuzz_parity.py, profiles/), every method, block splits, column
offsets, missing values, both routes of one-response NIPALS, new-data predict / transform -- no failure and every
attribute within 1e-7 of the oracle (tolerance 1e-6; the largest deviations, 1e-8 .. 3e-8, are SIMPLS with eight or nine
components, whose Gram-Schmidt recurrence :1015-1018 is the least well conditioned of the four methods)."""
import pytest

from tests import fuzz

pytestmark = pytest.mark.gpu


def test_random_problems(gpu_required):
    bad, worst = fuzz.run(150, seed=11, verbose=False)
    assert bad == 0
    assert worst < 1e-7, worst


def test_random_problems_register_resident_columns(gpu_required):
    """the same sweep on 1024 < n <= 8192 (ragged and odd sample counts, up to five responses)"""
    bad, worst = fuzz.run(15, seed=5, verbose=False, big=True)
    assert bad == 0
    assert worst < 1e-8, worst


def test_random_problems_long_columns(gpu_required):
    """... and on long columns (8192 < n <= 40000, narrow matrices): cluster passes, the storing two-phase passes, the
    multi-workgroup step kernel"""
    bad, worst = fuzz.run(16, seed=7, verbose=False, long=True)
    assert bad == 0
    assert worst < 1e-8, worst


def test_random_problems_wide(gpu_required):
    """... and on few samples x very many features (n <= 300, p up to 200 000): the column-group kernels at width"""
    bad, worst = fuzz.run(16, seed=9, verbose=False, wide=True)    # (the CPU oracle is the slow side at these widths)
    assert bad == 0
    assert worst < 1e-8, worst
