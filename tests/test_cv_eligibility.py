"""Which cross-validation jobs take the batched folds (mbpls_amd/cv_batch.eligible) -- the decision table, no GPU needed:
one response with every method (the same PLS1 model; SIMPLS only on standardised data, it centres its scores itself,
mbpls.py:1007-1009), several-response NIPALS, the p > n branches of several-response UNIPALS / KERNEL (:481-572, :694-805);
under column shards the branch is chosen from the TOTAL width."""
import numpy as np
import pytest

from mbpls_amd import MBPLS, cv_batch


class TwoRanks:
    """the other rank holds as many columns again"""
    is_distributed, world_size, rank = True, 2, 0

    def allreduce_host(self, a):
        return 2.0 * np.asarray(a, dtype=np.float64)


@pytest.mark.parametrize("method", ["NIPALS", "UNIPALS", "KERNEL", "SIMPLS"])
def test_one_response_every_method(method):
    assert cv_batch.eligible(MBPLS(n_components=3, method=method), 1, 100, 500, 10)
    assert cv_batch.eligible(MBPLS(n_components=3, method=method), 1, 500, 100, 10)      # n >= p: the same model still


def test_one_response_exceptions():
    assert not cv_batch.eligible(MBPLS(n_components=3, method="SIMPLS", standardize=False), 1, 100, 500, 10)
    assert cv_batch.eligible(MBPLS(n_components=3, method="KERNEL", standardize=False), 1, 100, 500, 10)
    assert not cv_batch.eligible(MBPLS(n_components=3, method="NIPALS", nipals_convergence_norm=1), 1, 100, 500, 10)
    assert not cv_batch.eligible(MBPLS(n_components=3, method="NIPALS", sparse_data=True), 1, 100, 500, 10)
    assert not cv_batch.eligible(MBPLS(n_components=33, method="NIPALS"), 1, 100, 500, 10)
    assert not cv_batch.eligible(MBPLS(n_components=3, method="NIPALS"), 1, 100, 500, 1)


def test_several_responses():
    assert cv_batch.eligible(MBPLS(n_components=3, method="NIPALS"), 4, 100, 500, 10)
    assert cv_batch.eligible(MBPLS(n_components=3, method="NIPALS"), 4, 500, 100, 10)
    assert not cv_batch.eligible(MBPLS(n_components=3, method="NIPALS"), cv_batch.MAX_RESPONSES + 1, 100, 500, 10)
    for method in ("UNIPALS", "KERNEL"):
        assert cv_batch.eligible(MBPLS(n_components=3, method=method), 4, 100, 500, 10)
        assert not cv_batch.eligible(MBPLS(n_components=3, method=method), 4, 500, 100, 10)     # n >= p branch: fold by fold
    assert not cv_batch.eligible(MBPLS(n_components=3, method="SIMPLS"), 4, 100, 500, 10)


def test_branch_follows_the_total_width_under_column_shards():
    est = MBPLS(n_components=3, method="UNIPALS")
    assert not cv_batch.eligible(est, 4, 100, 60, 10)                       # 60 columns, 100 samples: n >= p
    assert cv_batch.eligible(est, 4, 100, 60, 10, comm=TwoRanks())          # 120 columns over two ranks: p > n
