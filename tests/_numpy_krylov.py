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

    def set_vector(self, j, v):
        self.V[j] = v

    def get_vector(self, j):
        return self.V[j].copy()

    def step(self, j, apply=True):
        w = self.A @ self.V[j] if apply else self.V[j + 1].copy()
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
