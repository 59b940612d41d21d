"""Route of a fit: the switches that select among the solver routes, read ONCE per fit and in one place, and the record
of the route a fit actually took (``MBPLS.fit_plan_``).

The reference has one code path per method; this package has several per method (device-resident queue or host-driven
loop; lazy or explicit deflation; implicit, statistics-only or stored standardisation; for several responses the
inner loop in the space of its iterates, on association matrices, or sweeping X).  All of them are parity-tested
against each other and the oracle; the defaults below pick the fastest one that applies, the environment variables
exist for those tests and for A/B measurements.
"""
from __future__ import annotations

import dataclasses
import os


def _flag(name, default):
    v = os.environ.get(name)
    return default if v is None else v != "0"


@dataclasses.dataclass
class FitOptions:
    deflation: str | None = None        # MBPLS_DEFLATION = lazy | explicit: overrides the solver's own choice
    implicit_std: bool = True           # MBPLS_IMPLICIT_STD=0: always standardise into a stored working copy
    kernel_implicit: str | None = None  # MBPLS_KERNEL_IMPLICIT = 1 | 0: force statistics-only / stored preparation
    resident: bool = True               # MBPLS_RESIDENT=0: host-driven loop instead of the device-resident queue
    stream_stats: bool = True           # MBPLS_STREAM_STATS=0: scaler statistics fetched at the end of the fit
    nipals_subspace: bool = True        # MBPLS_NIPALS_SUBSPACE=0: several responses without the q-space inner loop
    nipals_qspace: bool = True          # MBPLS_NIPALS_QSPACE=0: that loop on n-vectors (the reference's own formulas) for the 2-norm too
    nipals_gram: bool = True            # MBPLS_NIPALS_GRAM=0: ... and without the association-matrix form either
    unipals_nodeflate: bool = True      # MBPLS_UNIPALS_NODEFLATE=0: UNIPALS, several responses, on the deflated X
    q1_fixed_point: bool = True         # MBPLS_NIPALS_Q1_FIXED_POINT=0: one-response NIPALS with the reference's two passes per component (the second only re-derives the first)
    r_incremental: bool = True          # MBPLS_R_INCREMENTAL=0: R_ / beta_ of a device-resident fit from one product at its end

    @classmethod
    def from_env(cls):
        d, k = os.environ.get("MBPLS_DEFLATION"), os.environ.get("MBPLS_KERNEL_IMPLICIT")
        return cls(deflation=d if d in ("lazy", "explicit") else None,
                   implicit_std=_flag("MBPLS_IMPLICIT_STD", True),
                   kernel_implicit=k if k in ("0", "1") else None,
                   resident=_flag("MBPLS_RESIDENT", True),
                   stream_stats=_flag("MBPLS_STREAM_STATS", True),
                   nipals_subspace=_flag("MBPLS_NIPALS_SUBSPACE", True),
                   nipals_qspace=_flag("MBPLS_NIPALS_QSPACE", True),
                   nipals_gram=_flag("MBPLS_NIPALS_GRAM", True),
                   unipals_nodeflate=_flag("MBPLS_UNIPALS_NODEFLATE", True),
                   q1_fixed_point=_flag("MBPLS_NIPALS_Q1_FIXED_POINT", True),
                   r_incremental=_flag("MBPLS_R_INCREMENTAL", True))
