"""turn a tools/run_prof.sh output directory into the committed artifacts:
profiles/<tag>_kernel_stats.csv, profiles/<tag>_pmc_summary.txt and
profiles/pmc_traffic.json (HBM bytes per launch, read by bench.py).

python tools/make_profiles.py gpurun_out/<dir> <tag> <nstamps>"""
import csv
import glob
import json
import os
import shutil
import sys
from collections import defaultdict

src, tag, nstamps = sys.argv[1], sys.argv[2], int(sys.argv[3])
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(root, "profiles")
os.makedirs(out, exist_ok=True)
shutil.copy(os.path.join(src, "stats", "run_kernel_stats.csv"),
            os.path.join(out, "%s_kernel_stats.csv" % tag))

acc = defaultdict(lambda: defaultdict(list))
c5acc = defaultdict(list)
for d in sorted(glob.glob(os.path.join(src, "pmc_*"))):
    f = os.path.join(d, "run_counter_collection.csv")
    if not os.path.exists(f):
        continue
    if os.path.basename(d).startswith("pmc_c5"):
        # config 5's loglike is the same kernel symbol as config 2's: kept apart
        for row in csv.DictReader(open(f)):
            if "pixpass_wave_kernel7<0" in row["Kernel_Name"] or \
                    "pixpass_wave_kernel<0" in row["Kernel_Name"]:
                c5acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
        continue
    team_dir = os.path.basename(d).startswith("pmc_team")
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0].replace("void ngmix::", "")
        # (the team passes run another workload -- tools/team_probe.py, 20,000
        # objects x 9 bands: only the team kernel is taken from them)
        if team_dir != k.startswith("lm_advance_team"):
            continue
        # (the traffic passes of config 3 bring FETCH_SIZE / WRITE_SIZE only: their
        # GRBM_GUI_ACTIVE belongs to other launches than the SQ passes' and must not
        # enter the issue figures' mean)
        if os.path.basename(d) in ("pmc_lmfetch", "pmc_lmwrite") and (
                row["Counter_Name"] not in ("FETCH_SIZE", "WRITE_SIZE") or "pixpass" in k):
            continue
        if "pixpass" in k or "admom" in k or "em_" in k or "lm_" in k:
            acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))

lines = ["# rocprofv3 --pmc passes (tools/run_prof.sh), mean per launch, "
         "%d stamps per launch" % nstamps]
traffic = {"nstamps": nstamps, "source": "profiles/%s_pmc_summary.txt" % tag,
           "method": "HBM bytes = 2*FETCH_SIZE*1024 + WRITE_SIZE*1024 per launch (median "
                     "over the launches of the PMC pass); "
                     "the factor 2 is MI355X_MICROARCH.md's gfx950 correction "
                     "(FETCH_SIZE = TCC_EA0_RDREQ x 64 B while the requests are "
                     "128 B), confirmed here: TCC_EA0_RDREQ_sum x 128 B equals the "
                     "algorithmic bytes of these kernels to 3%"}
for k, cs in sorted(acc.items()):
    for c, vals in sorted(cs.items()):
        lines.append("%-36s %-26s n=%d mean %.6g" % (k, c, len(vals),
                                                     sum(vals) / len(vals)))
    if "FETCH_SIZE" in cs and "WRITE_SIZE" in cs:
        # median, not mean: the workload builder's own fresh render (the
        # overwriting form, which never reads the image) is a launch of the
        # same symbol and must not dilute the accumulate-into launches
        fetch = sorted(cs["FETCH_SIZE"])[len(cs["FETCH_SIZE"]) // 2]
        write = sorted(cs["WRITE_SIZE"])[len(cs["WRITE_SIZE"]) // 2]
        name = {"pixpass_wave_kernel<0, false, 8>": "loglike",
                "pixpass_wave_kernel7<0, false, 8>": "loglike",
                "pixpass_wave_kernel<2, false, 16>": "render"}.get(k)
        if name:
            traffic[name + "_hbm_bytes_per_launch"] = 2 * fetch * 1024 + write * 1024
            traffic[name + "_fetch_size_kb"] = fetch
            traffic[name + "_write_size_kb"] = write
# config 3: the HBM bytes of the lm_eval launches of one complete fit (the FETCH_SIZE
# and WRITE_SIZE passes run the same deterministic program, tools/bench_lm.py: the
# i-th launch of one pass is the i-th of the other)
ev = acc.get("lm_eval_kernel<true, true>", {})
ninit = [len(cs["FETCH_SIZE"]) for k, cs in acc.items() if "lm_init_kernel" in k and
         "FETCH_SIZE" in cs]
if "FETCH_SIZE" in ev and "WRITE_SIZE" in ev and len(ev["FETCH_SIZE"]) == len(ev["WRITE_SIZE"]) \
        and ninit and ninit[0] > 0:
    per = [2 * f * 1024 + w * 1024 for f, w in zip(ev["FETCH_SIZE"], ev["WRITE_SIZE"])]
    traffic["c3_lm_eval_hbm_bytes_per_fit"] = sum(per) / ninit[0]
    traffic["c3_lm_eval_full_pass_hbm_bytes"] = max(per)
    traffic["c3_lm_eval_launches_per_fit"] = len(per) / float(ninit[0])
    traffic["c3_nstamps"] = nstamps
    lines.append("# config 3 (tools/bench_lm.py %d 0): lm_eval_kernel<true, true>, %d launches over "
                 "%d fits: HBM bytes per fit %.4g, of the fullest launch %.4g"
                 % (nstamps, len(per), ninit[0], sum(per) / ninit[0], max(per)))
# kernel-trace --stats of the bench command itself and of the other configs'
# drivers, trimmed to this library's kernels
if "FETCH_SIZE" in c5acc and "WRITE_SIZE" in c5acc:
    fetch = sum(c5acc["FETCH_SIZE"]) / len(c5acc["FETCH_SIZE"])
    write = sum(c5acc["WRITE_SIZE"]) / len(c5acc["WRITE_SIZE"])
    lines.append("# config 5 (bench.py --config C5: 20000 objects x 10 epochs of 64x64 per launch)")
    lines.append("%-36s %-26s n=%d mean %.6g" % ("pixpass_wave_kernel7<0, false, 8>",
                                                 "FETCH_SIZE", len(c5acc["FETCH_SIZE"]), fetch))
    lines.append("%-36s %-26s n=%d mean %.6g" % ("pixpass_wave_kernel7<0, false, 8>",
                                                 "WRITE_SIZE", len(c5acc["WRITE_SIZE"]), write))
    traffic["c5_loglike_hbm_bytes_per_launch"] = 2 * fetch * 1024 + write * 1024
    traffic["c5_nstamps"] = 200000
    for c, vals in sorted(c5acc.items()):
        if c not in ("FETCH_SIZE", "WRITE_SIZE"):
            lines.append("%-36s %-26s n=%d mean %.6g" % (
                "pixpass_wave_kernel7<0, false, 8>", c, len(vals), sum(vals) / len(vals)))
for sub, name in (("bench_stats", "bench"), ("c3_stats", "c3"), ("c4_stats", "c4"),
                  ("c5_stats", "c5"),
                  ("iter_stats", "iter"), ("lm_stats", "lm"), ("team_stats", "team")):
    f = os.path.join(src, sub, "run_kernel_stats.csv")
    if not os.path.exists(f):
        continue
    rows = list(csv.reader(open(f)))
    keep = [rows[0]] + [r for r in rows[1:] if "ngmix::" in r[0]]
    with open(os.path.join(out, "%s_%s_kernel_stats.csv" % (tag, name)), "w") as fo:
        csv.writer(fo, quoting=csv.QUOTE_ALL).writerows(keep)
    lines.append("# rocprofv3 --kernel-trace --stats -- %s (ngmix kernels): calls, average ns"
                 % {"bench": "python3 bench.py --no-cpu-baseline --no-other-configs",
                    "c3": "python3 bench.py --config C3 --steps 20 --warmup 2 --no-cpu-baseline",
                    "c4": "python3 bench.py --config C4 --steps 20 --warmup 5",
                    "c5": "python3 bench.py --config C5 --steps 50 --warmup 10",
                    "iter": "python3 tools/bench_iter.py 200000 3",
                    "lm": "python3 tools/bench_lm.py 100000 0",
                    "team": "python3 tools/lm_advance_share.py 10000"}[name])
    for r in keep[1:]:
        lines.append("%-60s calls %5s avg %12.0f ns" % (
            r[0].split("(")[0].replace("void ngmix::", "")[:60], r[1], float(r[3])))
for bname in ("bench", "bench_c3", "bench_c4", "bench_c5"):
    bj = os.path.join(src, bname + ".json")
    if os.path.exists(bj):
        shutil.copy(bj, os.path.join(out, "%s_%s.json" % (tag, bname)))
for logname in ("iter.log", "lm.log", "team.log"):
    f = os.path.join(src, logname)
    if os.path.exists(f):
        txt = [l for l in open(f).read().splitlines()
               if l.startswith(("admom", "em_run", "batched LM", "   mean", "exp x", "bdf x",
                                "coellip-"))]
        lines.append("# %s" % logname)
        lines.extend(txt)
# ---- instruction-issue figures of the VALU-bound kernels, read by bench.py
# next to each roofline block (as pmc_traffic.json is for the HBM bytes):
#   valu_busy            = SQ_ACTIVE_INST_VALU x 4 cycles / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs)
#   valu_insts_per_stamp = SQ_INSTS_VALU (wave instructions) of a launch with
#                          every stamp running / the stamps of that launch
def issue_figures(cs, n, busiest=False):
    if not all(k in cs for k in ("SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "GRBM_GUI_ACTIVE")):
        return None
    mean = lambda v: sum(v) / len(v)
    insts = max(cs["SQ_INSTS_VALU"]) if busiest else mean(cs["SQ_INSTS_VALU"])
    fig = {"valu_busy": mean(cs["SQ_ACTIVE_INST_VALU"]) * 4.0 /
                        (mean(cs["GRBM_GUI_ACTIVE"]) / 8.0 * 1024.0),
           "valu_insts_per_stamp": insts / n, "stamps_per_launch": n,
           "launches": len(cs["SQ_INSTS_VALU"])}
    if "SQ_INSTS_SALU" in cs:
        fig["salu_insts_per_stamp"] = (max(cs["SQ_INSTS_SALU"]) if busiest
                                       else mean(cs["SQ_INSTS_SALU"])) / n
    if "SQ_WAIT_INST_ANY" in cs and "SQ_WAVE_CYCLES" in cs:
        fig["wait_inst_frac_of_wave_cycles"] = mean(cs["SQ_WAIT_INST_ANY"]) / \
            mean(cs["SQ_WAVE_CYCLES"])
    return fig


valu = {"source": "profiles/%s_pmc_summary.txt" % tag,
        "method": "valu_busy = SQ_ACTIVE_INST_VALU x 4 / (GRBM_GUI_ACTIVE / 8 x 1024); "
                  "valu_insts_per_stamp = SQ_INSTS_VALU / stamps per launch (lm_eval: the "
                  "launch with every fit running and the full pass); rocprofv3 --pmc "
                  "passes of tools/run_prof.sh / run_prof_all.sh"}
for k, cs in sorted(acc.items()):
    # (the team form of lm_advance is profiled on tools/team_probe.py: 20,000
    # fits of 14 parameters per launch)
    n_k = 20000 if k.startswith("lm_advance_team") else nstamps
    fig = issue_figures(cs, n_k, busiest=k.startswith(("lm_eval", "lm_advance_team")))
    if fig:
        valu[k] = fig
fig = issue_figures(c5acc, 200000)
if fig:
    valu["c5:pixpass_wave_kernel7<0, false, 8>"] = fig
json.dump(valu, open(os.path.join(out, "pmc_valu.json"), "w"), indent=1)
open(os.path.join(out, "%s_pmc_summary.txt" % tag), "w").write("\n".join(lines) + "\n")
json.dump(traffic, open(os.path.join(out, "pmc_traffic.json"), "w"), indent=1)
print("\n".join(lines[:60]))
print(json.dumps(traffic, indent=1))
