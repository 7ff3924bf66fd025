"""numpy restatement of the control flow around the RMSD metric on enspara's
clustering hot path -- TEST INFRASTRUCTURE (see oracle/__init__.py).

Follows (file:line relative to /root/reference/enspara/cluster/):
  kcenters()                  kcenters.py:108-240  (loop :195-240)
  kcenters iteration          kcenters.py:243-311
  assign_to_nearest_center()  util.py:159-205
  find_cluster_centers()      util.py:208-242
  PAM sweep                   kmedoids.py:520-699 (+ :478-517)
  hybrid()                    hybrid.py:112-162

Data are float32 [n, A, 3] coordinate arrays; the metric is ``oracle.qcp``.
Results use the reference's dtypes (int64 assignments, float64 distances that
hold float32 values).  Pinned against the reference itself by
tests/golden/make_golden.py, which runs the real reference functions with
``oracle.qcp.rmsd`` plugged in as the metric.
"""
import numpy as np
from sklearn.utils import check_random_state

from . import qcp


def _metric_on(P):
    """metric(center_xyz) -> float32[n] against the prepared frames P."""
    def f(center_xyz):
        cc, Gc = qcp.center_and_trace(center_xyz)
        return qcp.rmsd_centered(P.c, P.G, cc[0], Gc[0])
    return f


def assign_to_nearest_center(X, centers):
    """util.py:199-203: one pass per center, strict < keeps the earlier center."""
    P = X if isinstance(X, qcp.Prepared) else qcp.Prepared(X)
    m = _metric_on(P)
    assignments = np.zeros(P.n, dtype=np.int64)
    distances = np.full(P.n, np.inf, dtype=np.float64)
    for i, ctr in enumerate(centers):
        d = m(ctr)
        closer = d < distances
        distances[closer] = d[closer]
        assignments[closer] = i
    return assignments, distances


def find_cluster_centers(assignments, distances):
    """util.py:233-242: per occupied label, first index of its minimum distance."""
    labels = np.unique(assignments)
    out = np.zeros_like(labels)
    for j, lab in enumerate(labels):
        members = np.flatnonzero(assignments == lab)
        out[j] = members[np.argmin(distances[members])]
    return out


def kcenters(X, n_clusters=None, dist_cutoff=None, init_centers=None):
    """kcenters.py:195-240 with the serial iteration of :243-311.
    Returns (center_indices list, assignments int64, distances float64)."""
    if n_clusters is None:
        n_clusters = np.inf
    if dist_cutoff is None:
        dist_cutoff = 0
    P = X if isinstance(X, qcp.Prepared) else qcp.Prepared(X)
    m = _metric_on(P)
    if init_centers is None:
        ctr_inds = []
        assignments = np.full(P.n, -1, dtype=np.int64)
        distances = np.full(P.n, np.inf, dtype=np.float64)
    else:
        assignments, distances = assign_to_nearest_center(P, init_centers)
        ctr_inds = list(find_cluster_centers(assignments, distances))
    maxdist = distances.max()
    while len(ctr_inds) < n_clusters and maxdist > dist_cutoff:
        new = int(np.argmax(distances))          # kcenters.py:282
        d = m(P.xyz[new])                        # :298
        closer = d < distances                   # :304
        distances[closer] = d[closer]
        assignments[closer] = len(ctr_inds)      # :306
        ctr_inds.append(new)
        maxdist = distances.max()                # :226
    return ctr_inds, assignments, distances


def msq(d):
    """kmedoids.py:478-479 at mpi size 1: mean of squares in float64."""
    return np.square(np.asarray(d, dtype=np.float64)).mean()


def pam_update(X, medoid_inds, assignments, distances, proposals=None,
               random_state=None, stop_after=None, budget_s=None, done=None):
    """One PAM sweep, kmedoids.py:575-699 (non-MPI branch), with the trial
    state of every proposal (:637-678: the proposal's distances, the three
    masks, the ambiguous members against all medoids) built by
    ``qcp_oracle.c:eko_pam_trial`` -- the loop ``pam_update_numpy`` below
    spells out in numpy, at a cost a complete 5000-proposal sweep over 10^6
    frames can afford (tests/test_oracle.py holds the two equal).  The draw
    (:514) and the cost comparison (:478-479, :680-683) are numpy's own.
    ``stop_after`` / ``budget_s`` / ``done`` as in ``pam_update_numpy``."""
    import time
    t_start = time.perf_counter()
    P = X if isinstance(X, qcp.Prepared) else qcp.Prepared(X)
    random_state = check_random_state(random_state)          # :579
    medoid_inds = list(medoid_inds)
    K = len(medoid_inds)
    med_c = np.ascontiguousarray(P.c[np.asarray(medoid_inds, dtype=np.int64)])
    med_G = np.ascontiguousarray(P.G[np.asarray(medoid_inds, dtype=np.int64)])
    # (copies: the two pairs of buffers swap roles when a proposal is accepted)
    distances = np.array(distances, dtype=np.float64)
    assignments = np.array(assignments, dtype=np.int64)
    new_dist = np.empty_like(distances)
    new_assig = np.empty_like(assignments)
    scratch = np.empty(P.n, dtype=np.float32)
    cost = msq(distances)       # the same call on the same array every time
    for cid in range(K if stop_after is None else min(stop_after, K)):
        if budget_s is not None and time.perf_counter() - t_start > budget_s:
            break
        if done is not None:
            done[:] = [cid + 1]
        if proposals is None:
            members = np.flatnonzero(assignments == cid)     # :611
            prop = random_state.choice(members)              # :514
        else:
            prop = proposals[cid]
        qcp.pam_trial(P, med_c, med_G, cid, P.c[prop], P.G[prop], distances,
                      assignments, new_dist, new_assig, scratch)
        new_cost = msq(new_dist)
        if new_cost < cost:                                  # :680-683
            cost = new_cost
            distances, new_dist = new_dist, distances
            assignments, new_assig = new_assig, assignments
            med_c[cid] = P.c[prop]
            med_G[cid] = P.G[prop]
            medoid_inds[cid] = prop
    return medoid_inds, distances, assignments


def pam_update_numpy(X, medoid_inds, assignments, distances, proposals=None,
                     random_state=None, stop_after=None, budget_s=None,
                     done=None):
    """One PAM sweep, kmedoids.py:575-699 (non-MPI branch).  ``stop_after`` /
    ``budget_s`` (checker only): only the first that many clusters' proposals,
    or as many as fit the seconds -- what a full-size cross-check can afford;
    ``done`` (a list) receives the number of proposals made."""
    import time
    t_start = time.perf_counter()
    P = X if isinstance(X, qcp.Prepared) else qcp.Prepared(X)
    m = _metric_on(P)
    random_state = check_random_state(random_state)          # :579
    medoid_inds = list(medoid_inds)
    medoid_xyz = [P.xyz[i] for i in medoid_inds]             # :607
    for cid in range(len(medoid_inds) if stop_after is None
                     else min(stop_after, len(medoid_inds))):
        if budget_s is not None and time.perf_counter() - t_start > budget_s:
            break
        if done is not None:
            done[:] = [cid + 1]
        members = np.flatnonzero(assignments == cid)         # :611
        if proposals is None:
            # the state was wrapped once per sweep (:579), so an int seed
            # restarts the stream every sweep while a RandomState instance
            # (what KHybrid passes, hybrid.py:78,103) continues it
            prop = random_state.choice(members)              # :514
        else:
            prop = proposals[cid]
        prop_xyz = P.xyz[prop]
        nd = m(prop_xyz)                                     # :637
        new_dist = np.zeros_like(distances) - 1
        new_assig = np.zeros_like(assignments) - 1
        down = distances > nd                                # :644
        new_assig[down] = cid
        new_dist[down] = nd[down]
        up_other = (distances <= nd) & (assignments != cid)  # :651
        new_assig[up_other] = assignments[up_other]
        new_dist[up_other] = distances[up_other]
        up_this = (distances <= nd) & (assignments == cid)   # :658
        trial = list(medoid_xyz)
        trial[cid] = prop_xyz
        sub = np.flatnonzero(up_this)
        if len(sub):
            sa, sd = assign_to_nearest_center(P.xyz[sub], trial)   # :666
            new_assig[sub] = sa
            new_dist[sub] = sd
        if msq(new_dist) < msq(distances):                   # :680-683
            distances, assignments = new_dist, new_assig
            medoid_xyz = trial
            medoid_inds[cid] = prop
    return medoid_inds, distances, assignments


def hybrid(X, n_clusters=None, dist_cutoff=None, n_iters=5, random_state=None,
           init_centers=None):
    """hybrid.py:112-162: k-centers, then n_iters PAM sweeps."""
    P = X if isinstance(X, qcp.Prepared) else qcp.Prepared(X)
    inds, assignments, distances = kcenters(
        P, n_clusters=n_clusters, dist_cutoff=dist_cutoff,
        init_centers=init_centers)
    for _ in range(n_iters):
        inds, distances, assignments = pam_update(
            P, inds, assignments, distances, random_state=random_state)
    return inds, assignments, distances
