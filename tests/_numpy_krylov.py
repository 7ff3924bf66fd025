"""numpy stand-in for enspara_amd.msm.transition_matrices.DeviceKrylov
(TEST CODE): same protocol, so the host side of the eigensolver can be
checked against scipy on a machine without a GPU."""
import numpy as np
import scipy.sparse


class NumpyKrylov:
    def __init__(self, A, m_max):
        self.A = scipy.sparse.csr_matrix(A).astype(np.float64)
        self.n = self.A.shape[0]
        self.m_max = m_max
        self.V = np.zeros((m_max + 1, self.n))
        self.filter = None

    def set_filter(self, degree, a=0.0, b=1.0):
        """the operator of step(): A, or T_degree((A - c) / e) on [a, b]"""
        self.filter = (int(degree), 0.5 * (a + b), 0.5 * (b - a)) if degree else None

    def _apply(self, x):
        if self.filter is None:
            return self.A @ x
        d, c, e = self.filter
        y0, y1 = x, (self.A @ x - c * x) / e
        for _ in range(2, d + 1):
            y0, y1 = y1, 2.0 * (self.A @ y1 - c * y1) / e - y0
        return y1

    def set_vector(self, j, v):
        self.V[j] = v

    def get_vector(self, j):
        return self.V[j].copy()

    def step(self, j, apply=True):
        w = self._apply(self.V[j]) if apply else self.V[j + 1].copy()
        h = np.zeros(j + 2)
        for _ in range(2):
            c = self.V[:j + 1] @ w
            w = w - c @ self.V[:j + 1]
            h[:j + 1] += c
        h[j + 1] = np.linalg.norm(w)
        if h[j + 1] > 0:
            self.V[j + 1] = w / h[j + 1]
        return h

    def rotate(self, m, Q, move_last):
        Q = np.asarray(Q)
        kk = Q.shape[1]
        new = Q.T @ self.V[:m]
        last = self.V[m].copy() if move_last else None
        self.V[:kk] = new
        if move_last:
            self.V[kk] = last

    def combine(self, m, Q):
        return np.asarray(Q).T @ self.V[:m]
