#!/bin/bash
# Where a 20-launch timed block of bench.py spends its time: rocprofv3 kernel trace of `bench.py --steps 20`, per-kernel durations and
# the gaps between consecutive kernels of a block (on the GPU box):  bash tools/trace20.sh
set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
D=$R/gpurun_out/trace20
rm -rf $D
rocprofv3 --kernel-trace --output-format csv -d $D -- python3 $R/bench.py --steps 20 --warmup 5 --repeats 30 --no-cpu-baseline --rollout-k 0 --image-envs 0 > $R/gpurun_out/trace20_bench.json 2>/dev/null
cd $R
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/trace20/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "sf_step_kernel" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# blocks: gaps > 50 us separate them
blocks = []; cur = [rows[0]]
for a, b in zip(rows, rows[1:]):
    if int(b["Start_Timestamp"]) - int(a["End_Timestamp"]) > 30000: blocks.append(cur); cur = []
    cur.append(b)
blocks.append(cur)
blocks = [b for b in blocks if len(b) == 20]
print("blocks of 20:", len(blocks))
import statistics as st
for name, b in (("median-ish block", blocks[len(blocks)//2]), ("last block", blocks[-1])):
    s = [int(r["Start_Timestamp"]) for r in b]; e = [int(r["End_Timestamp"]) for r in b]
    print(name, "span %.1f us" % ((e[-1] - s[0]) / 1e3), "durations us:", " ".join("%.1f" % ((y - x) / 1e3) for x, y in zip(s, e)))
    print("   gaps us:", " ".join("%.1f" % ((s[i + 1] - e[i]) / 1e3) for i in range(19)))
spans = [(int(b[-1]["End_Timestamp"]) - int(b[0]["Start_Timestamp"])) / 1e3 for b in blocks]
print("span of 20 kernels: median %.1f us min %.1f max %.1f" % (st.median(spans), min(spans), max(spans)))
PY
rm -rf $D
