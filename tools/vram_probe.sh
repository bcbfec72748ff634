#!/bin/bash
# tools/vram_probe.sh: tools/vram_probe.hip as the first GPU process of a box, again at once, then after pauses
hipcc --offload-arch=gfx950 -O2 -o /tmp/vram_probe tools/vram_probe.hip 2>/dev/null || exit 1
for pause in 0 0 0 2 4 8 0 0; do
  sleep $pause; echo -n "after ${pause}s: "; /tmp/vram_probe ${1:-56}
done
