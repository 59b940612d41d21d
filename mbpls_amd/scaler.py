"""Column scaler objects stored in ``x_scalers_`` / ``y_scaler_`` (reference mbpls.py:307,316,325).

The reference keeps sklearn ``StandardScaler`` instances there and users read ``mean_`` /
``scale_`` from them.  For X the statistics are produced by the HIP standardise pass; this class
only carries them (and can apply them to small host arrays such as Y).  ``fit`` implements the
same corrected two-pass algorithm for host data (used for Y, n x q)."""
from __future__ import annotations

import numpy as np

_EPS = np.finfo(np.float64).eps


class ColumnScaler:
    with_mean = True
    with_std = True

    def __init__(self, mean=None, scale=None, n_samples=None):
        if mean is not None:
            self.mean_ = np.asarray(mean, dtype=np.float64)
            self.scale_ = np.asarray(scale, dtype=np.float64)
            self.n_samples_seen_ = n_samples
            self.n_features_in_ = self.mean_.size

    def fit(self, X):
        X = np.asarray(X, dtype=np.float64)
        n = X.shape[0]
        self.mean_ = np.sum(X, axis=0) / n
        d = X - self.mean_
        corr = np.sum(d, axis=0)
        d **= 2
        ssq = np.sum(d, axis=0)
        ssq -= corr ** 2 / n
        self.var_ = ssq / n
        bound = n * _EPS * self.var_ + (n * self.mean_ * _EPS) ** 2
        self.scale_ = np.sqrt(self.var_)
        self.scale_[self.var_ <= bound] = 1.0
        self.n_samples_seen_ = n
        self.n_features_in_ = X.shape[1]
        return self

    def transform(self, X, copy=True):
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
