"""One screen of a bench.py line:  python3 tools/bench_digest.py gpurun_out/TAG_bench.json"""
import json
import sys
r = json.load(open(sys.argv[1]))
rf = r["roofline"]
print("value %.3e ms/step %.4f frac %.3f kernel_us %.2f" % (r["value"], r["ms_per_step"], rf["frac"], rf["kernel_us"]))
print("bound", rf.get("bound"), "| launch:", (r.get("config") or {}).get("launch", "")[:110])
if "single_launch" in rf:
    print("single launch: %.2f us frac %.3f plain %.2f us" % (rf["single_launch"]["kernel_us"], rf["single_launch"]["frac"], rf["single_launch"]["kernel_us_plain_launches"]))
if "valu_issue" in rf:
    v = rf["valu_issue"]
    print("valu_issue: busy %.2f us/step by counter = %.3f of the step (single launch %.3f); model frac %.3f" % (
        v["valu_busy_us_per_step_by_counter"], v["valu_busy_frac_by_counter"], v["valu_busy_frac_single_launch"], v["frac"]))
if "rollout" in r and "roofline" in r["rollout"]:
    x = r["rollout"]["roofline"]
    print("rollout_step_kernel: %.2f us frac %.3f valu-busy %s" % (x["kernel_us"], x["frac"], x.get("valu_busy_frac_by_counter")))
for k in ("roofline_f64_reward", "roofline_hbm_resident"):
    if k in r:
        print(k, "%.3f" % r[k]["frac"], "%.2f us" % r[k]["kernel_us"])
if "beam" in r:
    b = r["beam"]
    print("beam %.3e (best %.3e)  %.4f ms  valu-issue frac %s" % (b["value"], b["value_best_of_3_batches"], b["ms_per_batch_decision"],
                                                               (b.get("roofline") or {}).get("frac")))
print("in roofline:", {k: v for k, v in rf.items() if k.startswith(("beam_", "evaluation_"))})
print("in cpu_baseline:", {k: v for k, v in (r.get("cpu_baseline") or {}).items() if k.startswith(("beam_", "config1_", "value", "cores"))})
if "evaluation" in r:
    e = r["evaluation"]
    print("evaluation", e["seconds"], "no helpers", e["seconds_without_helper_wavefronts"], "same", e["same_games_without_helpers"],
          "with action stream", e.get("seconds_with_action_stream_and_best5_histories"), e.get("same_games_with_action_stream"))
if "rollout_random" in r:
    x = r["rollout_random"]
    print("rollout_random %.3e (%.2f us/step), bare %.3e, as launches %.3e" % (x["value"], x["us_per_step"], x["without_reward_stream"]["value"],
                                                                                 x["as_128_step_launches"]["value"]))
if "rollout" in r:
    print({k: v for k, v in r["rollout"].items() if isinstance(v, float)})
if "cpu_baseline_python" in r:
    print("config 1: drop-in %.0f steps/s, reference-style python env %.0f steps/s" % (r["cpu_baseline_python"].get("drop_in_steps_per_s", 0),
                                                                                         r["cpu_baseline_python"]["value"]))
print("headline", r.get("headline"))
