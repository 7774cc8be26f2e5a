#!/bin/bash
# A/B of the image step for libraries under build/abl (+ frames of the current build against round 3's library):
#   bash tools/r04_ab.sh LIB1 LIB2 ...     (GPU box; the current build = spacefortress_amd/libsfmi.so)
set -o pipefail
mkdir -p gpurun_out
SFMI_LIB_PATH=build/abl/libsfmi_r03.so python tools/render_hash.py gpurun_out/hash_r03.txt youturn 4096 1200 hunter > gpurun_out/hash.log 2>&1
python tools/render_hash.py gpurun_out/hash_new.txt youturn 4096 1200 hunter >> gpurun_out/hash.log 2>&1
cmp gpurun_out/hash_r03.txt gpurun_out/hash_new.txt && echo "FRAMES IDENTICAL to round 3 (youturn hunter, 4096 lanes x 1200 steps)" || echo "FRAMES DIFFER"
python tools/ab_render.py "$@" --rounds 3
