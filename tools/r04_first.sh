#!/bin/bash
# round 4, first GPU call: the image tests + the new record tests, frames of the new kernel against round 3's library, A/B timing
set -o pipefail
mkdir -p gpurun_out
python -m pytest tests/test_gpu_image.py tests/test_gpu_events.py -x -q -m gpu > gpurun_out/t_image.log 2>&1; echo "image tests rc=$?" | tee -a gpurun_out/summary.txt
tail -5 gpurun_out/t_image.log
SFMI_LIB_PATH=build/abl/libsfmi_r03.so python tools/render_hash.py gpurun_out/hash_r03.txt youturn 4096 1500 hunter > gpurun_out/hash.log 2>&1
python tools/render_hash.py gpurun_out/hash_new.txt youturn 4096 1500 hunter >> gpurun_out/hash.log 2>&1
cmp gpurun_out/hash_r03.txt gpurun_out/hash_new.txt && echo "FRAMES IDENTICAL youturn hunter" | tee -a gpurun_out/summary.txt || echo "FRAMES DIFFER youturn hunter" | tee -a gpurun_out/summary.txt
SFMI_LIB_PATH=build/abl/libsfmi_r03.so python tools/render_hash.py gpurun_out/hash_r03a.txt autoturn 4096 1500 hunter >> gpurun_out/hash.log 2>&1
python tools/render_hash.py gpurun_out/hash_newa.txt autoturn 4096 1500 hunter >> gpurun_out/hash.log 2>&1
cmp gpurun_out/hash_r03a.txt gpurun_out/hash_newa.txt && echo "FRAMES IDENTICAL autoturn hunter" | tee -a gpurun_out/summary.txt || echo "FRAMES DIFFER autoturn hunter" | tee -a gpurun_out/summary.txt
python tools/ab_render.py build/abl/libsfmi_r03.so spacefortress_amd/libsfmi.so --rounds 3 2>&1 | tee -a gpurun_out/summary.txt
for r in 1 2; do
  for l in build/abl/libsfmi_r03.so spacefortress_amd/libsfmi.so; do
    echo "$l: $(SFMI_LIB_PATH=$l python tools/image_probe.py 16384 300 stack 2>&1 | tail -1)" | tee -a gpurun_out/summary.txt
  done
done
