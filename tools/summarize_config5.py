"""BASELINE config 5 (one 16384 x 16384 frame, q = 10 / 50 / 90) from the rocprofv3 runs of `tools/gpu_run.sh config5 pmc_cal pmc_cal_wr`:
per quality the average launch time of dctq_strip_kernel (kernel trace + stats), its fraction of the 8 TB/s roofline at the
algorithmic 3 B/pixel (805.3 MB per launch), and the HBM-side traffic of one launch from the FETCH_SIZE / WRITE_SIZE passes, corrected
as MI355X_MICROARCH.md prescribes (KiB units; the read factor calibrated on tools/microbench's shape_io, whose bytes are known).

    python tools/summarize_config5.py r04      ->  profiles/r04_config5.csv, profiles/r04_config5.json (+ the three kernel-stats CSVs)
"""
import csv, glob, json, os, re, shutil, statistics, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
G, P = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles")
KERNEL = "dctq_strip_kernel"
ALG = 3.0 * 16384 * 16384


def newest(pattern):
    f = sorted(glob.glob(os.path.join(G, pattern)), key=os.path.getmtime, reverse=True)
    return f[0] if f else None


def counter(dirname, name, kernel_substr, grid=None):
    f = newest(os.path.join(dirname, "*", "*counter_collection.csv"))
    if not f:
        return None
    vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(f))
            if r["Counter_Name"] == name and kernel_substr in r["Kernel_Name"] and (grid is None or r["Grid_Size"] == grid)]
    return statistics.median(vals) if vals else None


cal_rd = counter("pmc_cal", "FETCH_SIZE", "shape_io", "2097152")
cal_wr = counter("pmc_cal_wr", "WRITE_SIZE", "shape_io", "2097152")
fr = (4096 * 4096 / 1024.0 / cal_rd) if cal_rd else 2.0
fw = (2 * 4096 * 4096 / 1024.0 / cal_wr) if cal_wr else 1.0
rows = []
for q in (10, 50, 90):
    row = {"quality": q}
    ks = newest("c5_prof_q%d/*/*kernel_stats.csv" % q)
    if ks:
        shutil.copy(ks, os.path.join(P, "%s_config5_q%d_kernel_stats.csv" % (tag, q)))
        for r in csv.DictReader(open(ks)):
            if KERNEL in r["Name"]:
                avg = float(r["AverageNs"])
                row.update(calls=int(r["Calls"]), average_ns=avg, min_ns=float(r["MinNs"]), max_ns=float(r["MaxNs"]),
                           achieved_GBps=round(ALG / avg, 1), frac_of_8TBps=round(ALG / avg / 8000.0, 4))
    kt = newest("c5_prof_q%d/*/*kernel_trace.csv" % q)
    txt = os.path.join(G, "c5_prof_q%d.txt" % q)
    line = None
    if os.path.exists(txt):
        for ln in open(txt).read().splitlines():
            if ln.startswith("{") and '"metric"' in ln:
                line = json.loads(ln)
    if kt and line:  # the K timed launches by their place in the trace: behind the line's untimed launches comes ONE submission of settling + W + K launches
        tr = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in csv.DictReader(open(kt)) if KERNEL in r["Kernel_Name"])
        sub = line["config"]["launches_in_the_timed_submission"]
        first, want, K = int(line["config"]["untimed_launches"]), sub["settling"] + sub["warmup"] + sub["timed"], sub["timed"]
        if len(tr) >= first + want:
            d = [e - s_ for s_, e in tr[first + want - K:first + want]]
            row.update(timed_steps=K, timed_steps_average_ns=round(statistics.mean(d), 1), timed_steps_frac_of_8TBps=round(ALG / statistics.mean(d) / 8000.0, 4))
    if line:
        row["bench_line_kernel_us_hip_events"] = float(line["roofline"]["kernel_us"])
    cold = os.path.join(G, "c5_cold_q%d.txt" % q)
    if os.path.exists(cold):  # the run without the profiler: its warm and cold figures (three rotating pairs, 2.4 GB)
        for ln in open(cold).read().splitlines():
            if ln.startswith("{") and '"metric"' in ln:
                cl = json.loads(ln)
                row.update(no_profiler_kernel_us=cl["roofline"]["kernel_us"], no_profiler_frac=cl["roofline"]["frac"], cold_kernel_us=cl["roofline"]["cold"]["kernel_us"],
                           cold_frac=cl["roofline"]["frac_hbm_cold"], parity=cl["config"]["parity"]["status"])
    rd, wr = counter("c5_rd_q%d" % q, "FETCH_SIZE", KERNEL), counter("c5_wr_q%d" % q, "WRITE_SIZE", KERNEL)
    if rd and wr:
        rb, wb = rd * 1024 * fr, wr * 1024 * fw
        row.update(FETCH_SIZE_KiB=rd, WRITE_SIZE_KiB=wr, hbm_read_bytes=round(rb), hbm_write_bytes=round(wb), hbm_bytes_per_launch=round(rb + wb),
                   algorithmic_bytes=ALG, traffic_over_algorithmic=round((rb + wb) / ALG, 4))
    rows.append(row)
out = {"round": tag, "workload": "BASELINE config 5: one 16384x16384 random uint8 frame (default_rng(1234)), resident in HBM, 805.3 MB algorithmic per launch",
       "command": "tools/gpu_run.sh config5 pmc_cal pmc_cal_wr (bench.py --height 16384 --width 16384 --quality q under rocprofv3 --kernel-trace --stats, "
                  "then --pmc FETCH_SIZE and --pmc WRITE_SIZE in passes of their own)",
       "calibration": {"fetch_factor": round(fr, 4), "write_factor": round(fw, 4), "FETCH_SIZE_reported_KiB": cal_rd, "WRITE_SIZE_reported_KiB": cal_wr,
                       "kernel": "tools/microbench.hip shape_io 4096x4096 (same access shape, known bytes)"},
       "reading": "average_ns is rocprofv3's average over ALL launches of the run (settling bursts of 256, the submission of 32 + 3 + 20 launches, the per-launch pass): "
                  "sustained back-to-back 800 MB launches run slower than the 20 timed launches (timed_steps_average_ns: those launches by their place in the same trace), "
                  "which is the interval bench.py's HIP events bracket (bench_line_kernel_us_hip_events, under the profiler; no_profiler_kernel_us without it); cold_* = three "
                  "rotating frame / coefficient pairs, 2.4 GB: nothing survives in the 256 MiB Infinity Cache - the honest HBM figure",
       "parity": "tests/test_gpu_parity.py::test_config5_16384_coefficient_digest: sha256 of the coefficients at q = 10, 50, 90 equals the reference's (manifest.json)",
       "rows": rows}
json.dump(out, open(os.path.join(P, "%s_config5.json" % tag), "w"), indent=1)
keys = ["quality", "calls", "average_ns", "min_ns", "max_ns", "achieved_GBps", "frac_of_8TBps", "timed_steps_average_ns", "timed_steps_frac_of_8TBps", "bench_line_kernel_us_hip_events", "no_profiler_kernel_us", "no_profiler_frac", "cold_kernel_us", "cold_frac", "parity", "hbm_read_bytes",
        "hbm_write_bytes", "hbm_bytes_per_launch", "traffic_over_algorithmic"]
with open(os.path.join(P, "%s_config5.csv" % tag), "w", newline="") as f:
    w = csv.DictWriter(f, fieldnames=keys, extrasaction="ignore")
    w.writeheader()
    for r in rows:
        w.writerow(r)
print(json.dumps(out, indent=1))
