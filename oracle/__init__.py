"""CPU oracle for the k-centers / RMSD hot path -- TEST INFRASTRUCTURE.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this package.  ``enspara_amd`` (the product) never does; it
fails loudly when its HIP library is missing instead of falling back to this.

``oracle.qcp``      ctypes binding of qcp_oracle.c (the RMSD arithmetic)
``oracle.cluster``  numpy restatement of the reference's control flow
                    (kcenters / assign / PAM), driven by ``oracle.qcp``
``oracle.msm``      numpy/scipy restatement of the MSM counting path
``oracle.xtc``      XTC decoder used to pin the RMSD against the reference's
                    mdtraj-produced known answers
"""
