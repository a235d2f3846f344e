import sys, numpy as np
sys.path.insert(0, "tests")
from cases import make_spec, problem_from_spec
from oracle_lib import oracle_eval
from smoothsde_amd import capi
for model, d in (("CTCRW", 7), ("CTCRW", 8), ("OU_SSM", 8)):
    for lengths in ([40, 7, 23, 2, 61, 1, 30], [5]):
        spec = make_spec("wide_coupled", model, d, seed=39, lengths=lengths, variant="const", na_rows=(), with_H=True)
        pb = problem_from_spec(spec)
        eng = capi.Engine(pb)
        v0 = eng.eval(spec["par"], order=0)
        v1, g1 = eng.eval(spec["par"], order=1)
        ov, og = oracle_eval(pb, spec["par"], order=1)
        print(model, d, lengths, "order0", v0 - ov, "order1", v1 - ov, "grad", np.max(np.abs(g1 - og)))
        eng.close()
