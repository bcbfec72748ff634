import os, subprocess, sys, time, json, tempfile
ROOT="/root/repo" if os.path.isdir("/root/repo/ska.rust_amd") else os.getcwd()
sys.path.insert(0, os.path.join(ROOT,"ska.rust_amd"))
import synth
mode=sys.argv[1]
if mode=="torch":
    import torch; print("avail", torch.cuda.is_available())
if mode=="torchimport":
    import torch
if mode=="hold":
    hold = bytearray(6*1024**3)
n=1000
td=tempfile.mkdtemp(dir="/dev/shm")
anc=synth.ancestor(5_000_000,seed=1)
files=[]
for i in range(n):
    p=os.path.join(td,f"g{i}.fa"); synth.to_fasta(synth.sample_stream(anc,i,n),p); files.append(p)
open(os.path.join(td,"list.txt"),"w").write("".join(f"g{i}\t{p}\n" for i,p in enumerate(files)))
SKA=os.path.join(ROOT,"ska.rust_amd","ska")
def run(args, ph):
    env=dict(os.environ, SKX_PHASES=os.path.join(td,ph))
    t=time.perf_counter(); r=subprocess.run([SKA,*args],cwd=td,capture_output=True,env=env); dt=time.perf_counter()-t
    assert r.returncode==0, r.stderr[-300:]
    return dt, json.load(open(os.path.join(td,ph)))
thr = "64"
tb,pb=run(["build","-f","list.txt","-o","all","-k","31","--threads",thr],"b.json")
if mode=="sleep": time.sleep(4)
align_args=["align","all.skf","-o","aln.fa"] + (["--threads",thr] if mode!="nothreads" else [])
ta,pa=run(align_args,"a.json")

print(mode, "build %.2f align %.2f" % (tb, ta))
print(" build:", {k: round(v, 3) for k, v in pb.items()})
print(" align:", {k: round(v, 3) for k, v in pa.items()})
import shutil; shutil.rmtree(td)
