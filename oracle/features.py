"""numpy restatement of the reference's native feature-space distances --
TEST INFRASTRUCTURE (see oracle/__init__.py).

Follows enspara/geometry/libdist.pyx: _euclidean :122-145, _manhattan
:100-117, _hamming :77-95, as compiled (float32 input: float32 difference and
float32 square, float64 running sum in feature order).  Pinned against outputs
of the reference's own compiled module (tests/golden/features_golden.npz)."""
import numpy as np


def _work(X, y):
    X = np.asarray(X)
    y = np.asarray(y)
    if X.dtype == np.float32:
        return X, y.astype(np.float32)
    return X.astype(np.float64), y.astype(np.float64)


def euclidean(X, y):
    X, y = _work(X, y)
    acc = np.zeros(len(X), dtype=np.float64)
    for j in range(X.shape[1]):
        d = X[:, j] - y[j]                 # in the working precision
        acc = acc + (d * d).astype(np.float64)
    return np.sqrt(acc)


def manhattan(X, y):
    X, y = _work(X, y)
    acc = np.zeros(len(X), dtype=np.float64)
    for j in range(X.shape[1]):
        acc = acc + np.abs((X[:, j] - y[j]).astype(np.float64))
    return acc


def hamming(X, y):
    X = np.asarray(X)
    y = np.asarray(y)
    acc = np.zeros(len(X), dtype=np.float64)
    for j in range(X.shape[1]):
        acc = acc + (X[:, j] != y[j])
    return acc / X.shape[1]
