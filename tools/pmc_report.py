#!/usr/bin/env python3
"""Turn two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; one counter per pass, as
MI355X_MICROARCH.md prescribes) of bench.py into HBM bytes per sf_step_kernel launch.

    SF_PMC_CALIB=1 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d DIR_F -- python bench.py ...
    SF_PMC_CALIB=1 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d DIR_W -- python bench.py ...
    python tools/pmc_report.py DIR_F DIR_W --envs 65536 --out profiles/rNN_pmc_traffic.json

gfx950 correction (guide, HBM section): FETCH_SIZE counts 64 B per 128-B request, i.e. half the
bytes of a coalesced stream.  It is calibrated here, not assumed: with SF_PMC_CALIB=1 bench.py
also dispatches sf_group_copy_kernel (sfmi.h: sf_calibration_copy) on groups of known size, in the
step kernel's own access pattern (16 bytes per lane, 64-lane rows); the read factor is fitted on
those, the write counter is checked to be exact on them.  Counter unit: KiB.
"""
import argparse
import collections
import csv
import glob
import json
import os
import re

CALIB_BYTES_PER_ENV = 20 * 16  # sf_group_copy_kernel: 20 slots x one 16-byte chunk


def load(d):
    f = glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True)[0]
    rows = list(csv.DictReader(open(f)))
    by = collections.defaultdict(list)
    for r in rows:
        by[r["Kernel_Name"], int(r.get("Grid_Size", 0) or 0)].append(float(r["Counter_Value"]))
    return by


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("fetch_dir")
    ap.add_argument("write_dir")
    ap.add_argument("--envs", type=int, default=65536)
    ap.add_argument("--gametype", default="youturn")
    ap.add_argument("--obs-type", default="features")
    ap.add_argument("--out")
    a = ap.parse_args()
    fe, wr = load(a.fetch_dir), load(a.write_dir)

    def calib(by):
        out = []
        for (k, _grid), v in by.items():
            if "sf_group_copy_kernel" in k:
                out += [("16B chunk rows", CALIB_BYTES_PER_ENV * a.envs, x * 1024) for x in v]
        return out

    cf, cw = calib(fe), calib(wr)
    read_factor = sum(b for _, b, m in cf) / sum(m for _, b, m in cf)
    write_factor = sum(b for _, b, m in cw) / sum(m for _, b, m in cw)
    lanes = (a.envs + 255) // 256 * 256  # only the launches of THIS workload (bench.py has legs at other sizes), one tick each
    one_tick = re.compile(r"sf_step_kernel<\w+, \w+, false\b")  # <AUTOTURN, SHAPED, FUSED = false, ...>
    pick = lambda by: [v for (k, grid), v in by.items() if one_tick.search(k) and grid == lanes][0]
    step_f, step_w = pick(fe), pick(wr)
    fetch_kib = sum(step_f) / len(step_f)
    write_kib = sum(step_w) / len(step_w)
    read_b = fetch_kib * 1024 * read_factor
    write_b = write_kib * 1024 * write_factor
    rep = {
        "workload": {"gametype": a.gametype, "envs_per_gpu": a.envs, "obs_type": a.obs_type},
        "kernel": "sf_step_kernel", "launches": len(step_f),
        "FETCH_SIZE_KiB_mean": fetch_kib, "WRITE_SIZE_KiB_mean": write_kib,
        "calibration": {"read_factor": read_factor, "write_factor": write_factor,
                        "read_samples": [dict(elem=t, true_bytes=b, counter_bytes=m) for t, b, m in cf],
                        "write_samples": [dict(elem=t, true_bytes=b, counter_bytes=m) for t, b, m in cw]},
        "read_bytes_per_launch": read_b, "write_bytes_per_launch": write_b,
        "traffic_bytes_per_launch": read_b + write_b,
        "per_env_step": {"read": read_b / a.envs, "write": write_b / a.envs},
    }
    print(json.dumps(rep, indent=1))
    if a.out:
        json.dump(rep, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
