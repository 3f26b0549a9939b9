#!/usr/bin/env python3
"""Fill the @PLACEHOLDER@ figures of a DESIGN.md template from a profile set:  python tools/fill_design.py <tag> [template] [out]
(figures come from profiles/r06_<tag>_*; INTEGRATION.md's @API_*@ figures are filled from the same set)."""
import csv
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
src = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "tools", "DESIGN.template.md")
dst = sys.argv[3] if len(sys.argv) > 3 else os.path.join(ROOT, "DESIGN.md")
P = os.path.join(ROOT, "profiles")


def L(name):
    return json.load(open(os.path.join(P, "r06_%s_bench%s.json" % (tag, name))))


def stats(name, kernel):
    """mean duration (us) of the first kernel whose name contains `kernel` in a rocprofv3 kernel_stats csv"""
    path = os.path.join(P, "r06_%s_%skernel_stats.csv" % (tag, name))
    for r in csv.DictReader(open(path)):
        if kernel in r["Name"]:
            return float(r["AverageNs"]) / 1e3
    return float("nan")


d = L("")
k = d["roofline"]["kernels"]
e = d["extra"]
c1, c3, c4s, c4, c5 = L("_cfg1"), L("_cfg3"), L("_cfg4_shard32"), L("_cfg4_256_rccl_1rank"), L("_cfg5")
k3 = c3["roofline"]["kernels"]
traffic = json.load(open(os.path.join(P, "r06_%s_pmc_traffic.json" % tag)))
t7, t15 = e["tracker_tree_sums"], c3["extra"]["tracker_tree_sums"]
pin = e["api_pinned_to_the_gpus_numa_node"]
gpu_tests = os.environ.get("KLT_NGPU", "189")
cpu_tests = os.environ.get("KLT_NCPU", "56")
v = {
    "CFG2_MS": "%.4f" % (d["ms_per_step"] / d["config"]["pairs_per_step"]),
    "CFG2_MF": "%.0f" % (d["value"] / 1e6),
    "SINGLE_MS": "%.4f" % d["single_pair"]["ms"], "SINGLE_MF": "%.0f" % (d["single_pair"]["features_per_s"] / 1e6),
    "CFG2_STEPFRAC": "%.2f" % d["roofline"]["step_frac"],
    "L0_US": "%.1f" % d["roofline"]["launch_us"], "L0_FRAC": "%.2f" % d["roofline"]["frac"],
    "L0_FRACA": "%.2f" % d["roofline"]["frac_vs_achievable"],
    "L0_ISSUE": "%.2f" % ((d["roofline"].get("issue_bound") or {}).get("frac") or float("nan")),
    "L0_PROF": "%.1f" % stats("", "smooth_grad_rb<unsigned char"),
    "L0_TRAFFIC": "%.1f" % ((d["roofline"]["traffic"] or float("nan")) / 1e6),
    "TAIL_US": "%.1f" % (k["pyramid_reduce"]["us_per_launch"] * k["pyramid_reduce"]["launches_per_step"] / k["smooth_grad_l0"]["launches_per_step"]
                         + k["gradients"]["us_per_launch"] * k["gradients"]["launches_per_step"] / k["smooth_grad_l0"]["launches_per_step"]),
    "TRK_US": "%.1f" % k["track"]["us_per_launch"], "TRK_FRAC": "%.2f" % k["track"]["frac"],
    "TRK_TRAFFIC": "%.1f" % (traffic.get("track", float("nan")) / 1e6),
    "PAIR_MB": "%.1f" % (d["roofline"]["step_algorithmic_bytes_formula"] / 1e6),
    "CFG1_MS": "%.4f" % c1["ms_per_step"],
    "CFG3_MS": "%.4f" % c3["ms_per_step"], "CFG3_TRK": "%.1f" % k3["track"]["us_per_launch"], "CFG3_AFF": "%.1f" % k3["affine_check"]["us_per_launch"],
    "CFG3_TRKF": "%.2f" % k3["track"]["frac"], "CFG3_AFFF": "%.2f" % k3["affine_check"]["frac"],
    "CFG4_MS": "%.2f" % c4["ms_per_step"], "CFG4_SHARD": "%.3f" % c4s["ms_per_step"], "CFG4_STEPFRAC": "%.2f" % c4["roofline"]["step_frac"],
    "CFG5_MS": "%.3f" % c5["ms_per_step"], "CFG5_MF": "%.0f" % (c5["value"] / 1e6), "CFG5_STEPFRAC": "%.2f" % c5["roofline"]["step_frac"],
    "SEL_MS": "%.3f" % e["ms_per_select_5000"],
    "PIPE_MS": "%.3f" % e["pcie_pipelined_ms_per_pair"], "PIPE_GB": "%.1f" % e["pcie_pipelined_GBps"], "PIPE_FRAC": "%.2f" % e["pcie_pipelined_frac_of_link"],
    "PIPE_FRACK": "%.2f" % e["pcie_pipelined_frac_of_link_next_to_a_kernel"],
    "SEQ1080": "%.3f" % e["sequence_from_host"]["1080p"]["ms_per_frame"], "SEQ4K": "%.3f" % e["sequence_from_host"]["4k"]["ms_per_frame"],
    "API_PP": "%.3f" % e["api_ms_per_KLTTrackFeatures_pingpong"], "API_SEL": "%.3f" % e["api_ms_per_KLTSelectGoodFeatures"],
    "API_CLIP": "%.3f" % e["api_ms_per_KLTTrackFeatures_consecutive_frames"], "API_FRESH": "%.3f" % e["api_ms_per_KLTTrackFeatures_new_frame_each_call"],
    "API_SEQLOOP": "%.3f" % e["api_ms_per_frame_sequential_mode_loop"], "TAG": tag,
    "RGB_PP": "%.3f" % e["api_ms_per_KLTTrackFeatures_pingpong_pil_rgb"], "RGB_FRESH": "%.3f" % e["api_ms_per_KLTTrackFeatures_new_frame_each_call_pil_rgb"],
    "RGB_SEL": "%.3f" % e["api_ms_per_KLTSelectGoodFeatures_pil_rgb"],
    "PIL_PP": "%.3f" % e["api_ms_per_KLTTrackFeatures_pingpong_pil"], "PIL_SEL": "%.3f" % e["api_ms_per_KLTSelectGoodFeatures_pil"],
    "PIL_CLIP": "%.3f" % e["api_ms_per_KLTTrackFeatures_consecutive_frames_pil"], "PIL_FRESH": "%.3f" % e["api_ms_per_KLTTrackFeatures_new_frame_each_call_pil"],
    "API_PP_P": "%.3f" % pin["api_ms_per_KLTTrackFeatures_pingpong"], "API_SEL_P": "%.3f" % pin["api_ms_per_KLTSelectGoodFeatures"],
    "API_CLIP_P": "%.3f" % pin["api_ms_per_KLTTrackFeatures_consecutive_frames"], "API_FRESH_P": "%.3f" % pin["api_ms_per_KLTTrackFeatures_new_frame_each_call"],
    "API_SEQLOOP_P": "%.3f" % pin["api_ms_per_frame_sequential_mode_loop"],
    "API_SEQ1080_P": "%.3f" % pin["api_ms_per_frame_KLTTrackSequence"]["1080p_5000_features_256_frames"],
    "API_SEQ4K_P": "%.3f" % pin["api_ms_per_frame_KLTTrackSequence"]["4k_20000_features_256_frames"],
    "API_SEQ1080": "%.3f" % e["api_ms_per_frame_KLTTrackSequence"]["1080p_5000_features_256_frames"],
    "API_SEQ4K": "%.3f" % e["api_ms_per_frame_KLTTrackSequence"]["4k_20000_features_256_frames"],
    "TREE7_US": "%.1f" % t7["us_per_launch"], "TREE7_EX": "%.1f" % t7["us_per_launch_exact"], "TREE7_X": "%.2f" % t7["speedup"],
    "TREE7_FRAC": "%.2f" % t7["frac"], "TREE7_DIFF": "%d" % t7["differing_positions"], "TREE7_DX": "%.3f" % t7["max_abs_dx"],
    "TREE15_US": "%.1f" % t15["us_per_launch"], "TREE15_EX": "%.1f" % t15["us_per_launch_exact"], "TREE15_X": "%.2f" % t15["speedup"],
    "TREE15_FRAC": "%.2f" % t15["frac"],
    "CPU_MS": "%.0f" % d["cpu_baseline"]["ms_per_pair"], "CPU_KF": "%.1f" % (d["cpu_baseline"]["value"] / 1e3),
    "NGPU": gpu_tests, "NCPU": cpu_tests,
}
text = open(src).read()
missing = sorted(set(re.findall(r"@([A-Z0-9_]+)@", text)) - set(v))
if missing:
    raise SystemExit("no value for " + ", ".join(missing))
for name, val in v.items():
    text = text.replace("@%s@" % name, val)
open(dst, "w").write(text)
