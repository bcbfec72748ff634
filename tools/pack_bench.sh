#!/bin/bash
# tools/pack_bench.sh [isolates]: the reader threads' work alone (tools/pack_bench.cpp) on freshly simulated read sets, one thread per file:
# framing only (0), framing + the byte-stream copy of round 3 (1), framing + bit-plane packing (2); then (2) with the knobs that take a piece away
n=${1:-8}; root=$(pwd); td=$(mktemp -d /dev/shm/pb.XXXX)
bin=$(mktemp /tmp/pack_bench.XXXX)    # (/dev/shm is mounted noexec on the GPU boxes)
g++ -O2 -pthread -o $bin tools/pack_bench.cpp -L ska.rust_amd -lskx -Wl,-rpath,$root/ska.rust_amd || exit 1
python - "$n" "$td" <<'PY'
import sys, os
sys.path.insert(0, "ska.rust_amd")
import synth
from concurrent.futures import ProcessPoolExecutor
n, td = int(sys.argv[1]), sys.argv[2]
with ProcessPoolExecutor(max_workers=min(n, 16)) as ex:
    list(ex.map(synth.write_read_pair_of, range(n), [n] * n, [os.path.join(td, f"iso{i}") for i in range(n)]))
PY
for rep in 1 2; do
  for m in 0 1 2; do $bin $m $td/*.fastq; done
  for k in simd_cap=2 simd_cap=1; do echo -n "$k: "; SKX_KNOBS=$k $bin 2 $td/*.fastq; done
  echo -n "one thread: "; $bin 2 $td/iso0_1.fastq; echo -n "one thread, framing only: "; $bin 0 $td/iso0_1.fastq
done
rm -rf $td $bin
