#!/usr/bin/env python3
"""Host. Turn one round's rocprofv3 passes (tools/profile_round.sh TAG -> gpurun_out/prof_TAG_*) into the files that are kept:

  profiles/TAG_pmc_step_beam_play_rollout.txt    the per-kernel summaries of the six counter passes (FETCH_SIZE, WRITE_SIZE, SQ_*
                                                 over bench.py's legs and over tools/rollout_rate.py 65536)
  profiles/TAG_bench_kernel_trace_stats.txt      --kernel-trace --stats of the driver's command + the chains timeline
  profiles/TAG_bench_under_tracer.json           the bench line of that traced run (slower than an untraced one: the tracer's
                                                 per-launch host cost; kept so that the trace can be read against its own line)
  profiles/pmc_step.json / pmc_beam.json / pmc_rollout.json
                                                 the counter values bench.py quotes (roofline.traffic, valu_issue): the numeric
                                                 fields are REPLACED by this round's averages, everything else is kept

usage: python tools/pmc_to_json.py TAG [gpurun_out]
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TAG = sys.argv[1] if len(sys.argv) > 1 else "r05"
OUT = sys.argv[2] if len(sys.argv) > 2 else os.path.join(REPO, "gpurun_out")
PROF = os.path.join(REPO, "profiles")


def newest(pattern):
    """gpurun merges a call's files into gpurun_out/: a pass directory can hold the files of earlier runs too. Only the newest set."""
    files = glob.glob(pattern, recursive=True)
    return sorted(files, key=os.path.getmtime)[-1:]


def counters(pass_name):
    """{kernel name: {counter: (n, mean)}} of one pass."""
    acc = defaultdict(lambda: defaultdict(list))
    for f in newest(os.path.join(OUT, "prof_%s_%s" % (TAG, pass_name), "**", "*counter_collection.csv")):
        for row in csv.DictReader(open(f)):
            acc[row["Kernel_Name"]][row["Counter_Name"]].append(float(row["Counter_Value"]))
    return {k: {c: (len(v), sum(v) / len(v)) for c, v in cs.items()} for k, cs in acc.items()}


def durations(pass_name, grid=None):
    by = defaultdict(list)
    for f in newest(os.path.join(OUT, "prof_%s_%s" % (TAG, pass_name), "**", "*kernel_trace.csv")):
        for row in csv.DictReader(open(f)):
            g = row.get("Grid_Size") or row.get("Grid_Size_X")
            if grid is None or str(grid) == g:
                by[row["Kernel_Name"]].append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
    return by


def pick(d, *needles):
    hits = [k for k in d if all(n in k for n in needles)]
    if len(hits) != 1:
        raise SystemExit("expected one kernel matching %r, found %r" % (needles, hits))
    return d[hits[0]]


def text(path):
    return open(os.path.join(OUT, path)).read()


def main():
    src = "profiles/%s_pmc_step_beam_play_rollout.txt" % TAG
    fetch, write, sq = counters("fetch"), counters("write"), counters("sq")
    rf, rw, rsq = counters("roll_fetch"), counters("roll_write"), counters("roll_sq")

    # ---- the kept text files
    with open(os.path.join(PROF, "%s_pmc_step_beam_play_rollout.txt" % TAG), "w") as o:
        o.write("# Round %s PMC passes (tools/profile_round.sh %s): separate rocprofv3 --pmc passes over `python3 bench.py --steps 20 --warmup 5 --chains 1`\n"
                "# (FETCH_SIZE; WRITE_SIZE; SQ_*: counters are per dispatch, so the one-launch-per-step form) and over `python3 tools/rollout_rate.py 65536`\n"
                "# (config 4's rollout_step_kernel). FETCH_SIZE / WRITE_SIZE in KB; gfx950: FETCH_SIZE x2 for wide coalesced reads (MI355X_MICROARCH.md).\n"
                "# SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_* in quad-cycles summed over waves. Reduced to profiles/pmc_*.json by tools/pmc_to_json.py.\n" % (str(int(TAG[1:])), TAG))
        for title, name in (("FETCH_SIZE pass, bench legs", "fetch"), ("WRITE_SIZE pass, bench legs", "write"), ("SQ pass, bench legs", "sq"),
                            ("FETCH_SIZE pass, rollout_step_kernel", "roll_fetch"), ("WRITE_SIZE pass, rollout_step_kernel", "roll_write"),
                            ("SQ pass, rollout_step_kernel", "roll_sq")):
            body = text("prof_%s_%s.summary.txt" % (TAG, name))
            keep = [ln for ln in body.splitlines() if "at::native" not in ln and "__amd_rocclr" not in ln and "Cijk_" not in ln and "attn_fwd" not in ln]
            o.write("\n## %s\n%s\n" % (title, "\n".join(keep).replace("(anonymous namespace)::", "")))
    with open(os.path.join(PROF, "%s_bench_kernel_trace_stats.txt" % TAG), "w") as o:
        o.write("# rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 20 --warmup 5   (round %s closing build; tools/profile_round.sh %s, pass kt)\n"
                "# The headline step of this run is the two-chain form: the tracer's per-launch host cost slows the host-paced launches (its own bench line:\n"
                "# profiles/%s_bench_under_tracer.json), so what the trace is good for is the per-kernel durations and the single-launch figure\n"
                "# (step_kernel<false,false,1,256> at grid 1048576), which the untraced run's roofline.single_launch reproduces.\n\n" % (str(int(TAG[1:])), TAG, TAG))
        o.write(text("prof_%s_kt.summary.txt" % TAG).replace("(anonymous namespace)::", ""))
        o.write("\n## step_kernel launches grouped by launch form (tools/chains_timeline.py)\n")
        o.write(text("prof_%s_kt.chains.txt" % TAG))
    line = [ln for ln in text("prof_%s_kt.json" % TAG).splitlines() if ln.startswith("{")][-1]
    json.dump(json.loads(line), open(os.path.join(PROF, "%s_bench_under_tracer.json" % TAG), "w"), indent=1)

    # ---- pmc_step.json
    def hbm(kernel_needles, boards):
        f, w = pick(fetch, *kernel_needles)["FETCH_SIZE"], pick(write, *kernel_needles)["WRITE_SIZE"]
        return f, w, int(round(2 * f[1] * 1024 + w[1] * 1024))

    p = os.path.join(PROF, "pmc_step.json")
    js = json.load(open(p))
    n = js["boards_per_launch"]
    head = ("step_kernel<false, false, 1, 256, false, false>",)
    f, w, total = hbm(head, n)
    s = pick(sq, *head)
    js.update({"FETCH_SIZE_KB": round(f[1], 1), "WRITE_SIZE_KB": round(w[1], 1), "hbm_bytes_per_launch": total,
               "ratio_to_algorithmic": total / js["algorithmic_bytes_per_launch"],
               "valu_wave_instructions_per_launch": int(round(s["SQ_INSTS_VALU"][1])), "waves": int(round(s["SQ_WAVES"][1])),
               "active_inst_valu_quad_cycles_per_launch": int(round(s["SQ_ACTIVE_INST_VALU"][1])),
               "sq_wave_cycles_quad_per_launch": int(round(s["SQ_WAVE_CYCLES"][1])),
               "sq_wait_inst_any_quad_per_launch": int(round(s["SQ_WAIT_INST_ANY"][1])),
               "source": "%s (round %s, separate --pmc passes, %d / %d launches)" % (src, str(int(TAG[1:])), f[0], w[0]),
               "sq_source": "%s (SQ pass, %d launches of step_kernel<false,false,1,256>)" % (src, s["SQ_INSTS_VALU"][0])})
    for key, needles in (("f64_reward_mode", ("step_kernel<true, false, 1, 256, false, false>",)),
                         ("hbm_resident_leg", ("step_kernel<false, false, 2, 256, false, false>",))):
        f2, w2, t2 = hbm(needles, js[key]["boards_per_launch"])
        js[key].update({"FETCH_SIZE_KB": round(f2[1], 1), "WRITE_SIZE_KB": round(w2[1], 1), "hbm_bytes_per_launch": t2,
                        "ratio_to_algorithmic": t2 / js[key]["algorithmic_bytes_per_launch"]})
    js["extra_legs_source"] = "%s (round %s: the same FETCH_SIZE / WRITE_SIZE passes cover bench.py's extra step legs)" % (src, str(int(TAG[1:])))
    json.dump(js, open(p, "w"), indent=1)

    # ---- pmc_beam.json
    p = os.path.join(PROF, "pmc_beam.json")
    jb = json.load(open(p))
    b = pick(sq, "beam_kernel<2>")
    jb.update({"valu_wave_instructions_per_launch": int(round(b["SQ_INSTS_VALU"][1])), "waves": int(round(b["SQ_WAVES"][1])),
               "valu_wave_instructions_per_expansion": b["SQ_INSTS_VALU"][1] / (jb["expansions_per_decision"] * b["SQ_WAVES"][1]),
               "active_inst_valu_quad_cycles_per_launch": int(round(b["SQ_ACTIVE_INST_VALU"][1])),
               "source": "%s (round %s, SQ_INSTS_VALU over %d launches)" % (src, str(int(TAG[1:])), b["SQ_INSTS_VALU"][0])})
    json.dump(jb, open(p, "w"), indent=1)

    # ---- pmc_rollout.json
    p = os.path.join(PROF, "pmc_rollout.json")
    jr = json.load(open(p))
    f, w, s = pick(rf, "rollout_step_kernel")["FETCH_SIZE"], pick(rw, "rollout_step_kernel")["WRITE_SIZE"], pick(rsq, "rollout_step_kernel")
    d = durations("roll_sq", 65536)
    dur = [x for k, v in d.items() if "rollout_step_kernel" in k for x in v]
    jr.update({"FETCH_SIZE_KB": round(f[1], 1), "WRITE_SIZE_KB": round(w[1], 1), "hbm_bytes_per_launch": int(round(2 * f[1] * 1024 + w[1] * 1024)),
               "valu_wave_instructions_per_launch": int(round(s["SQ_INSTS_VALU"][1])), "waves": int(round(s["SQ_WAVES"][1])),
               "wave_cycles_quad": int(round(s["SQ_WAVE_CYCLES"][1])),
               "wait_any_frac": round(s["SQ_WAIT_ANY"][1] / s["SQ_WAVE_CYCLES"][1], 3),
               "active_valu_frac": round(s["SQ_ACTIVE_INST_VALU"][1] / s["SQ_WAVE_CYCLES"][1], 3),
               "active_inst_valu_quad_cycles_per_launch": int(round(s["SQ_ACTIVE_INST_VALU"][1])),
               "kernel_us_rocprof_avg": round(sum(dur) / len(dur) / 1e3, 2), "source": src})
    json.dump(jr, open(p, "w"), indent=1)
    print(json.dumps({"step": {k: js[k] for k in ("hbm_bytes_per_launch", "ratio_to_algorithmic", "valu_wave_instructions_per_launch",
                                                  "active_inst_valu_quad_cycles_per_launch")},
                      "beam": jb["valu_wave_instructions_per_launch"], "rollout": {k: jr[k] for k in ("hbm_bytes_per_launch", "wait_any_frac", "active_valu_frac", "kernel_us_rocprof_avg")}}, indent=1))


if __name__ == "__main__":
    main()
