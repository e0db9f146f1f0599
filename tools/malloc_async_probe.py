"""Probe for the lost writes recorded in round 2: fixed-base multiplications of n scalars, both
groups, against the oracle; prints the first wrong index per case.  Run with PLAYSNARK_HIP_LIB pointing at a build with
-DPS_AFFINE_TMP_ASYNC (staging from hipMallocAsync) and at the shipped library."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import coracle as co, pyref as pr  # noqa: E402
from playsnark_amd import api  # noqa: E402

ctx = api.Context(0)
rng = pr.SplitMix64(300)
bad = 0
for rep in range(3):
    for og, gid in ((co.G1, api.G1), (co.G2, api.G2)):
        for n in (300, 255, 257, 1000, 5000):
            ks = [rng.fr() for _ in range(n)]
            got = api.Points.from_scalars(ctx, gid, api.Poly.upload(ctx, ks)).download()
            nb = og.nb
            wrong = [i for i in range(n) if got[i * nb:(i + 1) * nb] != og.to_b(og.mul(ks[i]))] if n <= 300 else None
            if wrong is None:  # spot check + identity count
                ident = sum(1 for i in range(n) if got[i * nb] == 0x40)
                wrong = [i for i in (0, 255, 256, 257, n - 1) if got[i * nb:(i + 1) * nb] != og.to_b(og.mul(ks[i]))]
                print(f"rep {rep} {og.name} n={n}: identities {ident}, wrong spot checks {wrong}")
            else:
                print(f"rep {rep} {og.name} n={n}: {len(wrong)} wrong, first {wrong[:3]}, last {wrong[-3:]}")
            bad += len(wrong)
print("TOTAL WRONG", bad)
