"""Route of a fit: the switches that select among the solver routes, read ONCE per fit and in one place, and the record
of the route a fit actually took (``MBPLS.fit_plan_``).

The reference has one code path per method; this package has several per method.  SIX of the choices are switches
(environment variables, below): each selects between two routes that both win somewhere, or restores the reference's
own structure for comparison.  The other forms a solver can take (``FALLBACK_FORMS``) are not switches: a fit takes
them when the default form does not cover its shape (more than 16 responses, a stop rule other than the 2-norm, an
engine without device helpers ...), and the tests force them through ``FitOptions.forced`` (tests/util.forced_routes)
to compare every form with the oracle -- they have lost every A/B since round 3 and nobody should select them.

| switch | default | the alternative |
|---|---|---|
| MBPLS_DEFLATION = lazy / explicit | lazy wherever it applies (one response, K < 39, no missing values, no row mask): X is never written | explicit: stored working copy, deflated in place (read + write) |
| MBPLS_IMPLICIT_STD = 0 | standardisation implicit in the passes (X never written) | stored standardised copy |
| MBPLS_KERNEL_IMPLICIT = 1 / 0 | KERNEL on the Gram: statistics-only preparation exactly when a working copy does not fit | force either |
| MBPLS_KERNEL_GRAM = 1 / 0 | KERNEL p > n: cost model (solvers._kernel_on_products) -- maintained thin products unless n < ~87 (K + 3) on fewer than 2^18 columns | force the matrix-core Gram / the products |
| MBPLS_RESIDENT = 0 | one response: the whole fit is one device queue | host-driven loop |
| MBPLS_NIPALS_Q1_FIXED_POINT = 0 | one-response NIPALS: one pass per component | the reference's two passes per component |
"""
from __future__ import annotations

import dataclasses
import os
from typing import ClassVar


def _flag(name, default):
    v = os.environ.get(name)
    return default if v is None else v != "0"


# forms a fit falls back to when the default one does not apply to its shape; forced only by the tests
FALLBACK_FORMS = ("stream_stats", "nipals_subspace", "nipals_qspace", "nipals_gram", "unipals_nodeflate", "r_incremental")


@dataclasses.dataclass
class FitOptions:
    deflation: str | None = None        # MBPLS_DEFLATION = lazy | explicit: overrides the solver's own choice
    implicit_std: bool = True           # MBPLS_IMPLICIT_STD=0: always standardise into a stored working copy
    kernel_implicit: str | None = None  # MBPLS_KERNEL_IMPLICIT = 1 | 0: force statistics-only / stored preparation
    kernel_gram: str | None = None      # MBPLS_KERNEL_GRAM = 1 | 0: KERNEL p > n on the matrix-core Gram / on maintained thin products (default: cost model)
    resident: bool = True               # MBPLS_RESIDENT=0: host-driven loop instead of the device-resident queue
    q1_fixed_point: bool = True         # MBPLS_NIPALS_Q1_FIXED_POINT=0: one-response NIPALS with the reference's two passes per component (the second only re-derives the first)
    # ---- FALLBACK_FORMS (no environment variable)
    stream_stats: bool = True           # False: scaler statistics fetched at the end of the fit
    nipals_subspace: bool = True        # False: several responses without the q-space inner loop (q > 16: association matrices)
    nipals_qspace: bool = True          # False: that loop on n-vectors (the reference's own formulas) for the 2-norm too (other norms always)
    nipals_gram: bool = True            # False: ... and without the association-matrix form either (engines without it)
    unipals_nodeflate: bool = True      # False: UNIPALS, several responses, on the explicitly deflated X
    r_incremental: bool = True          # False: R_ / beta_ of a device-resident fit from one product at its end (ill-conditioned P^T W)

    forced: ClassVar[dict] = {}         # tests only: {fallback form: value}, applied to every fit while set

    @classmethod
    def from_env(cls):
        d, k = os.environ.get("MBPLS_DEFLATION"), os.environ.get("MBPLS_KERNEL_IMPLICIT")
        kg = os.environ.get("MBPLS_KERNEL_GRAM")
        opts = cls(deflation=d if d in ("lazy", "explicit") else None,
                   implicit_std=_flag("MBPLS_IMPLICIT_STD", True),
                   kernel_implicit=k if k in ("0", "1") else None,
                   kernel_gram=kg if kg in ("0", "1") else None,
                   resident=_flag("MBPLS_RESIDENT", True),
                   q1_fixed_point=_flag("MBPLS_NIPALS_Q1_FIXED_POINT", True))
        for key, val in cls.forced.items():
            if key not in FALLBACK_FORMS:
                raise KeyError(f"{key} is not a fallback form ({FALLBACK_FORMS})")
            setattr(opts, key, bool(val))
        return opts
