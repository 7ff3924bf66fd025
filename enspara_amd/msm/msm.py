"""sklearn-style MSM wrapper (reference enspara/msm/msm.py:27-120; fit :60-88).
Ergodic trimming (trim=True) is not part of this build."""
import numpy as np

from ..exception import ImproperlyConfigured
from . import builders
from .transition_matrices import assigns_to_counts


class MSM(object):
    def __init__(self, lag_time, method, trim=False, sliding_window=True,
                 max_n_states=None, device=0):
        self.lag_time = lag_time
        self.trim = trim
        self.max_n_states = max_n_states
        self.method = method if callable(method) else getattr(builders, method)
        # the reference ignores its sliding_window argument (msm.py:58)
        self.sliding_window = True
        self.device = device

    @classmethod
    def from_assignments(cls, assignments, **kwargs):
        m = cls(**kwargs)
        m.fit(assignments)
        return m

    def fit(self, assigns):
        if self.trim:
            raise ImproperlyConfigured(
                "ergodic trimming is not available in this build")
        tcounts = assigns_to_counts(
            assigns, max_n_states=self.max_n_states, lag_time=self.lag_time,
            sliding_window=self.sliding_window, device=self.device)
        self.mapping_ = dict(zip(range(tcounts.shape[0]),
                                 range(tcounts.shape[0])))
        self.tcounts_, self.tprobs_, self.eq_probs_ = self.method(tcounts)
        return self

    @property
    def n_states_(self):
        if hasattr(self, "tprobs_"):
            return self.tprobs_.shape[0]
        raise ImproperlyConfigured(
            "MSM must be fit before it has a number of states.")
