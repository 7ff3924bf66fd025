import sys, time, numpy as np, scipy.sparse, cProfile, pstats
sys.path.insert(0, ".")
from enspara_amd.msm import eigenspectrum
K, L = 5000, 10000
n_trj = 300
rng = np.random.RandomState(11)
steps = rng.choice(np.array([-3,-2,-1,0,0,1,2,3],dtype=np.int8), size=(n_trj,L))
inblock = (rng.randint(100,size=(n_trj,1)) + np.cumsum(steps,axis=1,dtype=np.int32)) % 100
hops = np.cumsum(rng.rand(n_trj,L) < 0.002, axis=1, dtype=np.int32)
block = (rng.randint(K//100,size=(n_trj,1)) + hops*7) % (K//100)
A = (block*100+inblock).astype(np.int32)
rows=A[:,:-1].ravel(); cols=A[:,1:].ravel()
C = scipy.sparse.coo_matrix((np.ones(len(rows)),(rows,cols)),shape=(K,K)).tocsr()
T = scipy.sparse.diags(1.0/np.maximum(np.asarray(C.sum(1)).ravel(),1)) @ C
eigenspectrum(T, n_eigs=20)
pr=cProfile.Profile(); pr.enable(); eigenspectrum(T, n_eigs=20); pr.disable()
pstats.Stats(pr).sort_stats('tottime').print_stats(14)
