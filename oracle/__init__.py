"""TEST INFRASTRUCTURE ONLY: the CPU oracle (checker) for the playsnark hot path.

`oracle.coracle` = ctypes binding of oracle/liboracle.so (plain-C restatement),
`oracle.pyref`   = pure-Python big-int twin (small cases, fixture generation),
`oracle.restate` = Groth16Prove / PHGR13Prove composed from the C primitives, line by line
                   after groth16.go / pinochio.go.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this package.
"""
