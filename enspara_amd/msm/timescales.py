"""Implied timescales across lag times (reference enspara/msm/timescales.py:
calc_imp_times :12-42, implied_timescales :45-100), composed from the device
kernels: counts -> normalise -> leading eigenvalues, once per lag time."""
import numpy as np

from .transition_matrices import assigns_to_counts, eigenspectrum
from .trimming import trim_disconnected


def calc_imp_times(assigns, lag_time, n_states, n_times, method,
                   sliding_window, trim, device=0):
    C = assigns_to_counts(assigns, max_n_states=n_states, lag_time=lag_time,
                          sliding_window=sliding_window, device=device)
    if trim:                                              # timescales.py:26-27
        _, C = trim_disconnected(C)
    _, T, _ = method(C)
    n_times += 1                       # +1 accounts for the stationary mode
    e_vals, _ = eigenspectrum(T, n_eigs=n_times, device=device)
    return -lag_time / np.log(e_vals[1:])


def implied_timescales(assigns, lag_times, method, n_times=None,
                       sliding_window=True, trim=False, device=0):
    """-> array [len(lag_times), n_times]"""
    flat = assigns._data if hasattr(assigns, "_data") else np.asarray(assigns)
    n_states = int(flat.max()) + 1
    if n_times is None:
        n_times = int(np.floor(n_states / 10.0)) + 1
    if n_times > n_states - 1:
        n_times = n_states - 1
    return np.array([
        calc_imp_times(assigns, t, n_states, n_times, method, sliding_window,
                       trim, device=device) for t in lag_times])
