import sys
sys.path.insert(0,'/root/repo/ska.rust_amd'); sys.path.insert(0,'/root/repo/tests')
import numpy as np, skx_engine as E, ora
E.load_library()
rng=np.random.default_rng(1)
for k,L,nrec in ((31,3000,1),(31,2000,1),(31,4000,1),(21,6000,1),(41,1500,1),(41,2500,1)):
    recs=[bytes(rng.choice(np.frombuffer(b"ACGT",dtype=np.uint8),size=L).tolist()) for _ in range(nrec)]
    ds=E.DictSet.build([E.record_stream(recs)],k,False)
    gk,gb=ds.export(0)
    d=ora.Dict.new(k,False)
    for r in recs: d.add_record(r)
    ok,ob=d.export()
    G={(int(a['hi'])<<64)|int(a['lo']) for a in gk}; O={(int(a['hi'])<<64)|int(a['lo']) for a in ok}
    print("k",k,L,nrec,len(gk),len(ok)," common",len(G&O)," only gpu",len(G-O)," only ora",len(O-G))
