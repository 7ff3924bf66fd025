"""sklearn-style MSM wrapper (reference enspara/msm/msm.py:27-120; fit :60-88;
load/save :190-281)."""
import json
import os
import pickle
import shutil
import tempfile

import numpy as np
from scipy.io import mmread, mmwrite

from ..exception import ImproperlyConfigured
from . import builders
from .transition_matrices import assigns_to_counts
from .trimming import TrimMapping, trim_disconnected


class MSM(object):
    def __init__(self, lag_time, method, trim=False, sliding_window=True,
                 max_n_states=None, device=0):
        self.lag_time = lag_time
        self.trim = trim
        self.max_n_states = max_n_states
        if callable(method):
            self.method = method
        elif hasattr(builders, str(method)):
            self.method = getattr(builders, method)
        else:
            # reference builders: normalize, transpose, mle (Prinz's iteration,
            # builders.py:24 + libmsm.pyx: a sequential dense sweep that is not
            # part of this build's path)
            raise NotImplementedError(
                "MSM builder '%s' is not available here; supported: "
                "'normalize', 'transpose', or any callable "
                "counts -> (counts, probs, eq_probs)" % (method,))
        # the reference ignores its sliding_window argument (msm.py:58)
        self.sliding_window = True
        self.device = device

    @classmethod
    def from_assignments(cls, assignments, **kwargs):
        m = cls(**kwargs)
        m.fit(assignments)
        return m

    def fit(self, assigns):
        tcounts = assigns_to_counts(
            assigns, max_n_states=self.max_n_states, lag_time=self.lag_time,
            sliding_window=self.sliding_window, device=self.device)
        if self.trim:                                        # msm.py:74-79
            self.mapping_, tcounts = trim_disconnected(tcounts)
        else:
            self.mapping_ = TrimMapping(zip(range(tcounts.shape[0]),
                                            range(tcounts.shape[0])))
        self.tcounts_, self.tprobs_, self.eq_probs_ = self.method(tcounts)
        return self

    @classmethod
    def load(cls, path, manifest="manifest.json"):
        """Read back a directory written by ``save`` (reference msm.py:190-221):
        the manifest names the files holding the mapping (csv), counts and
        probabilities (MatrixMarket), equilibrium populations (text) and the
        pickled configuration."""
        if not os.path.isdir(path):
            raise NotImplementedError(
                "%s is not a directory; archived models are not supported" % path)
        with open(os.path.join(path, manifest)) as f:
            names = json.load(f)
        names = {k: os.path.join(path, v) for k, v in names.items()}
        with open(names["config"], "rb") as f:
            config = pickle.load(f)
        m = cls(**config)
        m.tcounts_ = mmread(names["tcounts_"])
        m.tprobs_ = mmread(names["tprobs_"])
        m.mapping_ = TrimMapping.load(names["mapping_"])
        m.eq_probs_ = np.loadtxt(names["eq_probs_"])
        return m

    def save(self, path, force=False, zipfile=False, **filenames):
        """Write the fitted model as a directory (reference msm.py:223-281;
        same file names, formats and manifest).  ``force`` replaces an existing
        directory (the reference calls ``os.remove`` on it, which cannot
        succeed; here the directory is removed)."""
        names = {"mapping_": "mapping.csv", "tcounts_": "tcounts.mtx",
                 "tprobs_": "tprobs.mtx", "eq_probs_": "eq-probs.dat",
                 "config": "config.pkl"}
        names.update(filenames)
        if zipfile:
            raise NotImplementedError("archived models are not supported")
        with tempfile.TemporaryDirectory(prefix=os.path.basename(path)) as tmp:
            with open(os.path.join(tmp, "manifest.json"), "w") as f:
                json.dump(names, f, sort_keys=True, indent=4,
                          separators=(",", ": "))
            with open(os.path.join(tmp, names["mapping_"]), "w") as f:
                self.mapping_.write(f)
            with open(os.path.join(tmp, names["tcounts_"]), "wb") as f:
                mmwrite(f, self.tcounts_)
            with open(os.path.join(tmp, names["tprobs_"]), "wb") as f:
                # 20 digits: the probabilities must survive the round trip
                mmwrite(f, self.tprobs_, precision=20)
            with open(os.path.join(tmp, names["eq_probs_"]), "wb") as f:
                np.savetxt(f, np.array(self.eq_probs_))
            with open(os.path.join(tmp, names["config"]), "wb") as f:
                pickle.dump(self.config, f)
            if force and os.path.isdir(path):
                shutil.rmtree(path)
            shutil.copytree(tmp, path)

    @property
    def config(self):
        """reference msm.py:101-111"""
        return {"lag_time": self.lag_time,
                "sliding_window": self.sliding_window,
                "trim": self.trim, "method": self.method}

    @property
    def result_(self):
        """reference msm.py:113-134 (None before fit)"""
        if getattr(self, "tcounts_", None) is None:
            return None
        return {"tcounts_": self.tcounts_, "tprobs_": self.tprobs_,
                "eq_probs_": self.eq_probs_, "mapping_": self.mapping_}

    def __eq__(self, other):
        """reference msm.py:136-177: same configuration and, if fit, the same
        counts, probabilities, equilibrium populations and mapping."""
        if self is other:
            return True
        if not isinstance(other, MSM) or self.config != other.config:
            return False
        if self.result_ is None or other.result_ is None:
            return self.result_ is None and other.result_ is None
        if not np.all(self.eq_probs_ == other.eq_probs_):
            return False
        if self.mapping_ != other.mapping_:
            return False
        if (self.tcounts_.shape != other.tcounts_.shape or
                self.tprobs_.shape != other.tprobs_.shape):
            return False
        if (self.tcounts_ != other.tcounts_).nnz != 0:
            return False
        return (self.tprobs_ != other.tprobs_).nnz == 0

    __hash__ = None

    def __repr__(self):
        return "MSM:" + str({"config": self.config, "fit": self.result_})

    @property
    def n_states_(self):
        if hasattr(self, "tprobs_"):
            return self.tprobs_.shape[0]
        raise ImproperlyConfigured(
            "MSM must be fit before it has a number of states.")
