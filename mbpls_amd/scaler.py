"""Column scaler objects stored in ``x_scalers_`` / ``y_scaler_`` (reference mbpls.py:307,316,325).

The reference keeps sklearn ``StandardScaler`` instances there and users read ``mean_`` /
``scale_`` from them.  For X the statistics are produced by the HIP standardise pass; this class
only carries them (and can apply them to small host arrays such as Y).  ``fit`` implements the
same corrected two-pass algorithm for host data (used for Y, n x q)."""
from __future__ import annotations

import numpy as np

try:        # the reference stores sklearn StandardScaler objects: be one (isinstance checks, sklearn's own tooling)
    from sklearn.preprocessing import StandardScaler as _Base
except Exception:  # pragma: no cover
    _Base = object

_EPS = np.finfo(np.float64).eps


class ColumnScaler(_Base):
    """A ``sklearn.preprocessing.StandardScaler`` (when scikit-learn is importable) whose fitted statistics were
    produced elsewhere -- for X by the HIP standardise pass -- with lean ``fit`` / ``transform`` for small host arrays."""

    def __init__(self, mean=None, scale=None, n_samples=None, var=None, *, copy=True, with_mean=True, with_std=True):
        self.copy, self.with_mean, self.with_std = copy, with_mean, with_std      # StandardScaler's parameters
        self._set(mean, scale, n_samples, var)

    def get_params(self, deep=True):
        return {"copy": self.copy, "with_mean": self.with_mean, "with_std": self.with_std}

    def _set(self, mean, scale, n_samples, var):
        """the fitted attributes of sklearn's StandardScaler: ``mean_``, ``scale_``, ``var_`` (0 for a constant
        column, whose ``scale_`` is 1), ``n_samples_seen_`` (an int, or one count per column when the data had
        missing values), ``n_features_in_``"""
        if mean is not None:
            self.mean_ = np.asarray(mean, dtype=np.float64)
            self.scale_ = np.asarray(scale, dtype=np.float64)
            if var is not None:
                self.var_ = np.asarray(var, dtype=np.float64)
            self.n_samples_seen_ = n_samples
            self.n_features_in_ = self.mean_.size

    def fit(self, X, y=None, sample_weight=None):
        """mean, then corrected two-pass variance (sklearn ``_incremental_mean_and_var``); NaN entries are ignored, as
        sklearn does (the statistics of a column are taken over its present samples)."""
        X = np.asarray(X, dtype=np.float64)
        miss = np.isnan(X)
        has_nan = bool(miss.any())
        cnt = (X.shape[0] - miss.sum(axis=0)).astype(np.float64) if has_nan else float(X.shape[0])
        self.mean_ = (np.nansum(X, axis=0) if has_nan else np.sum(X, axis=0)) / cnt
        d = X - self.mean_
        if has_nan:
            d[miss] = 0.0
        corr = np.sum(d, axis=0)
        d **= 2
        ssq = np.sum(d, axis=0)
        ssq -= corr ** 2 / cnt
        self.var_ = ssq / cnt
        bound = cnt * _EPS * self.var_ + (cnt * self.mean_ * _EPS) ** 2
        self.scale_ = np.sqrt(self.var_)
        self.scale_[self.var_ <= bound] = 1.0
        self.n_samples_seen_ = cnt.astype(np.int64) if has_nan else X.shape[0]
        self.n_features_in_ = X.shape[1]
        return self

    def transform(self, X, copy=True):
        return self._standardised(X)

    def _standardised(self, X):
        """``transform`` without sklearn's ``set_output`` wrapper around it (12 us per call; MBPLS.fit's own use)"""
        Z = np.array(X, dtype=np.float64, copy=True)
        Z -= self.mean_
        Z /= self.scale_
        return Z

    def fit_transform(self, X):
        return self.fit(X).transform(X)

    def inverse_transform(self, Z, copy=True):
        X = np.array(Z, dtype=np.float64, copy=True)
        X *= self.scale_
        X += self.mean_
        return X
