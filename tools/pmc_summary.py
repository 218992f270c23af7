#!/usr/bin/env python3
"""Turn two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; --output-format csv) into the per-launch HBM traffic
record bench.py reports as roofline.traffic.
usage: tools/pmc_summary.py <fetch counter_collection.csv> <write counter_collection.csv> <kernel substr> <bench.json of the same
       command (for bytes_per_launch / schedule)> <out.json>"""
import json
import sys

import pandas as pd


def main():
    fetch_csv, write_csv, kern, bench_json, out = sys.argv[1:6]
    bench = json.loads([l for l in open(bench_json) if l.startswith("{")][-1])
    roof = bench["roofline"]
    res = {}
    for kind, path in (("fetch", fetch_csv), ("write", write_csv)):
        df = pd.read_csv(path)
        k = df[df.Kernel_Name.str.contains(kern, regex=False)]
        res[kind] = dict(launches=int(len(k)), kb=float(k.Counter_Value.mean()),
                         us=float((k.End_Timestamp - k.Start_Timestamp).mean() / 1e3))
    fetch_b = res["fetch"]["kb"] * 1024 * 2      # gfx950: FETCH_SIZE tallies 16 B/lane coalesced reads at half (MI355X_MICROARCH.md, HBM)
    write_b = res["write"]["kb"] * 1024          # WRITE_SIZE taken as is (uncalibrated)
    if "bytes_per_update_algorithmic" in roof:   # round 3 lines: bytes_per_launch is the schedule-derived figure
        tuples = roof["bytes_per_epoch"] / roof["bytes_per_update"]
        alg = float(roof["bytes_per_update_algorithmic"]) * tuples / roof["launches_per_epoch"]
        model_b = float(roof["bytes_per_launch"])
    else:
        alg, model_b = float(roof["bytes_per_launch"]), None
    rec = {"kernel": kern, "schedule": roof.get("schedule", "level"), "workload": bench["config"]["workload"],
           "launches_profiled": res["fetch"]["launches"], "launches_per_epoch": roof["launches_per_epoch"],
           "fetch_bytes_per_launch": fetch_b, "write_bytes_per_launch": write_b,
           "hbm_bytes_per_launch": fetch_b + write_b, "algorithmic_bytes_per_launch": alg,
           "traffic_over_algorithmic": (fetch_b + write_b) / alg,
           "schedule_model_bytes_per_launch": model_b, "traffic_over_schedule_model": ((fetch_b + write_b) / model_b) if model_b else None,
           "profiled_launch_us": 0.5 * (res["fetch"]["us"] + res["write"]["us"]),
           "real_traffic_GBps_profiled": (fetch_b + write_b) / (0.5 * (res["fetch"]["us"] + res["write"]["us"])) / 1e3,
           "method": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes with --kernel-trace; "
                     "FETCH_SIZE KB x1024 x2 (gfx950 correction), WRITE_SIZE KB x1024"}
    json.dump(rec, open(out, "w"), indent=1)
    print(json.dumps(rec, indent=1))


if __name__ == "__main__":
    main()
