#!/bin/bash
# tools/prof_align.sh [n]: rocprofv3 kernel summary of `ska align x.skf` (and of `ska build`) on n synthetic 5 Mbp assemblies
n=${1:-1000}
R=$(pwd)
td=$(mktemp -d -p /dev/shm)
python - "$n" "$td" <<'PY'
import os, sys
sys.path.insert(0, "ska.rust_amd")
import synth
n = int(sys.argv[1]); td = sys.argv[2]
anc = synth.ancestor(5_000_000, seed=1)
with open(os.path.join(td, "list.txt"), "w") as f:
    for i in range(n):
        p = os.path.join(td, f"g{i}.fa"); synth.to_fasta(synth.sample_stream(anc, i, n), p); f.write(f"g{i}\t{p}\n")
PY
cd /tmp && export TMPDIR=/tmp
(cd $td && rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_build -- $R/ska.rust_amd/ska build -f list.txt -o all -k 31 --threads 64) > /dev/null 2>&1
sleep 3
(cd $td && SKX_PHASES=$td/ph.json rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_align -- $R/ska.rust_amd/ska align all.skf -o aln.fa --threads 64) > /dev/null 2>&1
cd $R
cat $td/ph.json; echo
for w in build align; do echo "== ska $w"; python tools/rocprof_stats.py $(ls gpurun_out/prof_$w/*/*.db | head -1) | head -14 | cut -c1-150; done
rm -rf $td gpurun_out/prof_build gpurun_out/prof_align
