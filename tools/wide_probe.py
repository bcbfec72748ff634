import sys, os
sys.path.insert(0, "tests"); sys.path.insert(0, "ska.rust_amd")
import numpy as np, ora, skx_engine as E
E.load_library(); E.default_context()
rng = np.random.default_rng(5)
for k in (33, 41, 47, 55):
    bad = []
    for n in (1500, 3000, 5000, 7000, 9000, 12000, 17000, 25000, 40000, 70000):
        rec = bytes(rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=n).tolist())
        ds = E.DictSet.build([E.record_stream([rec])], k, True)
        gk, gb = ds.export(0)
        d = ora.Dict.new(k, True); d.add_record(rec); ok, ob = d.export()
        if not (len(gk) == len(ok) and np.array_equal(gk["lo"], ok["lo"]) and np.array_equal(gk["hi"], ok["hi"]) and np.array_equal(gb, ob)):
            bad.append(n)
    print("k", k, "bad sizes", bad)
