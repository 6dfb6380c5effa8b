"""Turns the raw rocprofv3 output merged into gpurun_out/ into the small, committed summaries under profiles/.

    python tools/summarize_profiles.py r01        # round tag

Reads gpurun_out/prof (kernel trace + stats, warm: one frame replayed), gpurun_out/prof_cold (the same with the 12 rotating buffer
pairs making up 97 % of the launches), gpurun_out/pmc_rd / pmc_wr (FETCH_SIZE / WRITE_SIZE of bench.py) and
gpurun_out/pmc_cal / pmc_cal_wr (same counters on tools/microbench's shape_io kernel, whose byte counts are known).
HBM bytes follow MI355X_MICROARCH.md's recipe: counters are in KiB; on gfx950 FETCH_SIZE reads exactly half the
bytes of this kernel's read pattern (calibrated on shape_io: 8,204 KiB reported for 16,384 KiB read), WRITE_SIZE is
exact (32,768 KiB reported for 32 MiB written).
"""
import csv
import glob
import json
import os
import shutil
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
KERNEL = "dctq_strip_kernel"  # the production kernel (round 1: dctq_hybrid_kernel<0>)
G = os.path.join(ROOT, "gpurun_out")
P = os.path.join(ROOT, "profiles")
os.makedirs(P, exist_ok=True)


def counter(dirname, name, kernel_substr, grid=None):
    files = sorted(glob.glob(os.path.join(G, dirname, "*", "*counter_collection.csv")), key=os.path.getmtime, reverse=True)
    if not files:
        return None
    vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(files[0]))
            if r["Counter_Name"] == name and kernel_substr in r["Kernel_Name"] and (grid is None or r["Grid_Size"] == grid)]
    return statistics.median(vals) if vals else None


out = {"round": tag, "units": "KiB as reported by rocprofv3; bytes after correction"}
ks = sorted(glob.glob(os.path.join(G, "prof", "*", "*kernel_stats.csv")), key=os.path.getmtime, reverse=True)
if ks:
    shutil.copy(ks[0], os.path.join(P, f"{tag}_rocprofv3_kernel_stats.csv"))
    for r in csv.DictReader(open(ks[0])):
        if KERNEL in r["Name"]:
            out["kernel_stats"] = {"kernel": r["Name"], "calls": int(r["Calls"]), "average_ns": float(r["AverageNs"]),
                                   "min_ns": float(r["MinNs"]), "max_ns": float(r["MaxNs"])}
# The K timed launches: bench.py queues [settling launches of the submission, W warm-up launches, event, K timed launches, event] as ONE
# submission (config.launches_in_the_timed_submission); in the trace that is the first run of exactly that many back-to-back launches (gaps
# below 20 us) - the settling bursts in front of it are runs of 256, the per-launch pass behind it carries a 5 us gap behind every launch but
# may have the same count: the first such run is the timed one.  The average of its last K launches is the figure to hold against the same
# run's bench line (roofline.kernel_us = ms_per_step), on the same box; rocprofv3's own AverageNs includes every other launch of the run.
kt = sorted(glob.glob(os.path.join(G, "prof", "*", "*kernel_trace.csv")), key=os.path.getmtime, reverse=True)
bench_line = None
for cand in ("prof.txt",):
    pth = os.path.join(G, cand)
    if os.path.exists(pth):
        for ln in open(pth).read().splitlines():
            if ln.startswith("{") and '"metric"' in ln:
                bench_line = json.loads(ln)
if kt and bench_line:
    K = int(bench_line["steps"])
    sub = bench_line["config"].get("launches_in_the_timed_submission", {"settling": 0, "warmup": int(bench_line["warmup"]), "timed": K})
    want = int(sub["settling"]) + int(sub["warmup"]) + int(sub["timed"])
    rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in csv.DictReader(open(kt[0])) if KERNEL in r["Kernel_Name"]))
    # (the launches in front of the submission are counted by the bench line itself: config.untimed_launches = the settling bursts + the
    #  statistics launch; under the profiler a submission does not stay gap-free, so the launches are told by their place in the trace)
    first = int(bench_line["config"]["untimed_launches"])
    seg = (first, first + want) if len(rows) >= first + want else None
    if seg is not None:
        timed = rows[seg[1] - K:seg[1]]
        d = [e - s_ for s_, e in timed]
        t = statistics.mean(d)
        span = (timed[-1][1] - timed[0][0]) / K
        out["timed_steps"] = {"steps": K, "average_ns": round(t, 1), "frac_of_8TBps": round(3.0 * 4096 * 4096 / t / 8000.0, 4),
                              "first_start_to_last_end_per_launch_ns": round(span, 1), "min_ns": min(d), "max_ns": max(d),
                              "bench_line_kernel_us_hip_events": bench_line["roofline"]["kernel_us"], "bench_line_ms_per_step": bench_line["ms_per_step"],
                              "kernel_time_le_ms_per_step": bool(t / 1e6 <= bench_line["ms_per_step"] * 1.0005),
                              "note": "the K timed launches of the same rocprofv3 trace the bench line was printed under (launches %d .. %d of the trace: the last K of the submission of %d"
                                      " launches that follows the untimed ones): kernel time (trace) <= ms_per_step (HIP events around the K launches, which add what the queue needs between two launches)" % (seg[1] - K, seg[1] - 1, want)}
        with open(os.path.join(P, f"{tag}_bench_under_rocprof.json"), "w") as f:
            json.dump(bench_line, f)
    else:
        out["timed_steps"] = {"error": "the trace holds %d launches, the timed submission would end at launch %d" % (len(rows), first + want)}
kc = sorted(glob.glob(os.path.join(G, "prof_cold", "*", "*kernel_stats.csv")), key=os.path.getmtime, reverse=True)
if kc:
    shutil.copy(kc[0], os.path.join(P, f"{tag}_rocprofv3_kernel_stats_cold.csv"))
    for r in csv.DictReader(open(kc[0])):
        if KERNEL in r["Name"]:
            out["kernel_stats_cold"] = {"kernel": r["Name"], "calls": int(r["Calls"]), "average_ns": float(r["AverageNs"]),
                                        "min_ns": float(r["MinNs"]), "max_ns": float(r["MaxNs"]),
                                        "note": "bench.py --steps 50 --warmup 10 --settle-ms 1: ~2,500 of the launches rotate over 12 frame/coefficient "
                                                "buffer pairs (604 MB), ~70 replay one pair"}
cal_rd = counter("pmc_cal", "FETCH_SIZE", "shape_io", "2097152")
cal_wr = counter("pmc_cal_wr", "WRITE_SIZE", "shape_io", "2097152")
known_rd, known_wr = 4096 * 4096 / 1024.0, 2 * 4096 * 4096 / 1024.0
fr = (known_rd / cal_rd) if cal_rd else 2.0
fw = (known_wr / cal_wr) if cal_wr else 1.0
out["calibration"] = {"kernel": "tools/microbench.hip shape_io 4096x4096 (same access shape, known bytes)",
                      "FETCH_SIZE_reported_KiB": cal_rd, "read_KiB_actual": known_rd, "fetch_factor": round(fr, 4),
                      "WRITE_SIZE_reported_KiB": cal_wr, "written_KiB_actual": known_wr, "write_factor": round(fw, 4)}
rd = counter("pmc_rd", "FETCH_SIZE", KERNEL)
wr = counter("pmc_wr", "WRITE_SIZE", KERNEL)
if rd and wr:
    rb, wb = rd * 1024 * fr, wr * 1024 * fw
    out["hybrid_4096x4096_q50"] = {"FETCH_SIZE_KiB": rd, "WRITE_SIZE_KiB": wr, "hbm_read_bytes": rb, "hbm_write_bytes": wb,
                                    "hbm_bytes_per_launch": rb + wb, "algorithmic_bytes": 3.0 * 4096 * 4096,
                                    "traffic_over_algorithmic": round((rb + wb) / (3.0 * 4096 * 4096), 4)}
    json.dump({"hbm_bytes_per_launch": rb + wb, "source": f"profiles/{tag}_pmc_traffic.json"},
              open(os.path.join(P, "traffic_latest.json"), "w"))
json.dump(out, open(os.path.join(P, f"{tag}_pmc_traffic.json"), "w"), indent=1)
for name in ("bench.txt", "bench_cold.txt"):
    src = os.path.join(G, name)
    if os.path.exists(src):
        shutil.copy(src, os.path.join(P, f"{tag}_{name}"))
print(json.dumps(out, indent=1))
