#!/usr/bin/env python3
"""profiles/r06_driver_flags.txt from the `tools/gpu_run.sh driver_flags driver_prof` runs merged into gpurun_out/ (one lease per TIC_TAG):
the driver's command line against the long interval, run by run, and the timed launches of the rocprofv3 trace one by one."""
import csv, glob, json, os, re, sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = []
tags = sorted({re.search(r"df_tool_(\w+)\.txt", f).group(1) for f in glob.glob(os.path.join(root, "gpurun_out", "df_tool_*.txt"))})
out.append("# Round 6: `python3 bench.py --gpus 1 --steps 20 --warmup 5` (the driver's command) against `--steps 5000 --warmup 2000`, one gpurun lease per tag.")
out.append("# ms_per_step of the bench line: W warm-up launches, event, K timed launches, event in ONE submission (tic_dctq_dev_timed_warm), 256 settling launches in front.")
out.append("# kernel = the strip kernel's own duration by the dispatch packet's time stamps (per_launch_us, a pass of its own).")
for tag in tags:
    out.append("")
    out.append("## lease %s" % tag)
    rows = []
    for f in sorted(glob.glob(os.path.join(root, "gpurun_out", "df_short_%s*.txt" % tag))) + sorted(glob.glob(os.path.join(root, "gpurun_out", "df_long_%s.txt" % tag))):
        try:
            d = json.loads(open(f).read().strip().splitlines()[-1])
        except Exception as ex:  # noqa: BLE001
            out.append("%s: unreadable (%s)" % (os.path.basename(f), ex))
            continue
        pl = d["config"].get("per_launch_us") or {}
        note = ""
        if str(pl.get("source", "")).rstrip().endswith("the timed launches themselves"):
            note = "  [events on the timed launches' own packets: a 5 us gap behind every launch - NOT a valid line, kept as the measurement of that gap]"
        rows.append((d["steps"], d["ms_per_step"] * 1e3))
        out.append("%-18s steps %4d warmup %4d  ms_per_step %7.3f us  frac %.4f  cold %.4f  kernel min/median/max %s/%s/%s us  wall/launch %.3f us%s"
                   % (os.path.basename(f), d["steps"], d["warmup"], d["ms_per_step"] * 1e3, d["roofline"]["frac"], d["roofline"].get("frac_hbm_cold", float("nan")),
                      pl.get("min"), pl.get("median"), pl.get("max"), d["config"]["wall_ms_per_step"] * 1e3, note))
    tool = os.path.join(root, "gpurun_out", "df_tool_%s.txt" % tag)
    if os.path.exists(tool):
        keep = [l.rstrip() for l in open(tool) if re.match(r"^(idle|warm|preroll|steps)\s", l)]
        out.append("tools/driver_flags.py in one process (5 rounds, 60 ms of settling in front of every measurement):")
        out += ["  " + l for l in keep]
prof = sorted(glob.glob(os.path.join(root, "gpurun_out", "df_prof", "*", "*_kernel_trace.csv")), key=os.path.getmtime, reverse=True)
line = None
pth = os.path.join(root, "gpurun_out", "df_prof.txt")
if os.path.exists(pth):
    for ln in open(pth).read().splitlines():
        if ln.startswith("{") and '"metric"' in ln:
            line = json.loads(ln)
if prof and line:
    rows = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in csv.DictReader(open(prof[0])) if "strip" in r["Kernel_Name"])
    sub = line["config"]["launches_in_the_timed_submission"]
    want, first, K = sub["settling"] + sub["warmup"] + sub["timed"], int(line["config"]["untimed_launches"]), sub["timed"]
    out.append("")
    out.append("## the same command under rocprofv3 --kernel-trace (tools/gpu_run.sh driver_prof): the timed launches one by one")
    out.append("bench line of that run: ms_per_step %.3f us (HIP events), per_launch_us %s" % (line["ms_per_step"] * 1e3, {k: v for k, v in line["config"]["per_launch_us"].items() if k != "source"}))
    if len(rows) >= first + want:
        seg = rows[first:first + want]
        timed = seg[-K:]
        out.append("launches %d .. %d of the trace = the submission of %d settling + %d warm-up + %d timed launches; the %d timed ones, duration us / gap to the launch in front us:"
                   % (first, first + want - 1, sub["settling"], sub["warmup"], K, K))
        out.append("  " + "  ".join("%.2f/%.2f" % ((e - s) / 1e3, (s - seg[-K - 1 + k][1]) / 1e3) for k, (s, e) in enumerate(timed)))
        out.append("  first timed start .. last timed end: %.2f us = %.3f us per launch; mean duration %.3f us; the %d untimed launches of the submission in front: mean %.3f us"
                   % ((timed[-1][1] - timed[0][0]) / 1e3, (timed[-1][1] - timed[0][0]) / 1e3 / K, sum(e - s for s, e in timed) / 1e3 / K, want - K,
                      sum(e - s for s, e in seg[:-K]) / 1e3 / max(want - K, 1)))
    else:
        out.append("(the trace holds %d launches, the submission would end at %d)" % (len(rows), first + want))
open(os.path.join(root, "profiles", "r06_driver_flags.txt"), "w").write("\n".join(out) + "\n")
print("\n".join(out))
