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
prof = glob.glob(os.path.join(root, "gpurun_out", "df_prof", "*", "*_kernel_trace.csv"))
if prof:
    rows = [r for r in csv.DictReader(open(prof[0])) if "strip" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    # the bench's launches in order: 1 statistics launch, settling bursts, then ONE submission of 256 + 5 + 20 launches (the timed pass), later the per-launch pass
    ts = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows]
    gaps = [ts[i][0] - ts[i - 1][1] for i in range(1, len(ts))]
    # the timed submission = the first run of exactly 281 back-to-back launches behind a gap of more than 20 us
    starts = [0] + [i for i in range(1, len(ts)) if gaps[i - 1] > 20000]
    out.append("")
    out.append("## the same command under rocprofv3 --kernel-trace (gpurun_out/df_prof): the 20 timed launches one by one")
    for a, b in zip(starts, starts[1:] + [len(ts)]):
        if b - a == 57:
            seg = ts[a:b]
            timed = seg[-20:]
            out.append("submission of %d launches (32 settling + 5 warm-up + 20 timed); the 20 timed ones: duration us / gap to the launch in front us" % (b - a))
            out.append("  " + "  ".join("%.2f/%.2f" % ((e - s) / 1e3, (s - seg[-21 + k][1]) / 1e3) for k, (s, e) in enumerate(timed)))
            out.append("  first timed start .. last timed end: %.2f us = %.3f us per launch; mean duration %.3f us" % ((timed[-1][1] - timed[0][0]) / 1e3, (timed[-1][1] - timed[0][0]) / 2e4,
                                                                                                                        sum(e - s for s, e in timed) / 2e4))
            break
    else:
        out.append("(no submission of 57 launches found in the trace: %d strip-kernel launches, segments %s)" % (len(ts), [b - a for a, b in zip(starts, starts[1:] + [len(ts)])][:12]))
open(os.path.join(root, "profiles", "r06_driver_flags.txt"), "w").write("\n".join(out) + "\n")
print("\n".join(out))
