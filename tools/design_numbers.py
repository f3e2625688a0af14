"""Regenerate the measured numbers of DESIGN.md from the committed records under profiles/, so that the text cannot drift from them
("What's weak" #8).  Everything between the two marker lines of DESIGN.md is replaced.

    python tools/design_numbers.py [round tag, default r05]      (run it after copying new records into profiles/)

Inputs: profiles/<tag>_bench_line.json (one line of `python bench.py`), <tag>_kernel_steady.txt (tools/steady_profile.py),
<tag>_pmc_traffic.json (tools/pmc_traffic.py), <tag>_truth_tests.jsonl (tests/test_gpu_truth.py), <tag>_pmc_mfma.jsonl (tools/pmc_mfma.sh).
"""
import json
import os
import re
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TAG = sys.argv[1] if len(sys.argv) > 1 else "r05"
BEGIN, END = "<!-- BEGIN GENERATED: measurements (tools/design_numbers.py) -->", "<!-- END GENERATED -->"


def load_json(name):
    path = os.path.join(REPO, "profiles", name)
    if not os.path.exists(path):
        return None
    with open(path) as f:
        txt = f.read().strip()
    return json.loads(txt.splitlines()[-1]) if name.endswith("bench_line.json") else json.loads(txt)


def steady_table():
    path = os.path.join(REPO, "profiles", f"{TAG}_kernel_steady.txt")
    if not os.path.exists(path):
        return None, []
    lines = open(path).read().splitlines()
    rows = []
    for ln in lines[1:]:
        m = re.match(r"(.+?)\s+([\d.]+)/step\s+([\d.]+) us/step avg\s+([\d.]+) us\s+([\d.]+)%", ln)
        if m:
            name = re.sub(r"\(.*", "", m.group(1)).replace("void ", "").replace("lhg::", "").strip()
            rows.append((name, float(m.group(2)), float(m.group(3))))
    return lines[0], rows


def family_ms(rows):
    fam = {"gather-GEMM (gg*)": 0.0, "weight-gradient GEMM (wg*)": 0.0, "weight-gradient slab reduce": 0.0, "per-channel max (wgrad scales)": 0.0,
           "BatchNorm family (bn_*, reduce_partials)": 0.0, "thin convolutions": 0.0, "absmax (tensor scales)": 0.0, "weight packing": 0.0,
           "angular-spectrum passes": 0.0, "other": 0.0}
    launches = dict.fromkeys(fam, 0.0)
    for name, n, us in rows:
        if "wgrad_reduce" in name or "wg6_reduce" in name:
            k = "weight-gradient slab reduce"
        elif "channel_absmax" in name:
            k = "per-channel max (wgrad scales)"
        elif "thin_" in name:
            k = "thin convolutions"
        elif re.search(r"gg\d*[a-z]*_kernel", name):
            k = "gather-GEMM (gg*)"
        elif re.search(r"(?<![a-z])wg\d*[a-z]*_kernel", name):
            k = "weight-gradient GEMM (wg*)"
        elif name.startswith("bn_") or "reduce_partials" in name:
            k = "BatchNorm family (bn_*, reduce_partials)"
        elif "absmax" in name:
            k = "absmax (tensor scales)"
        elif "pack_" in name:
            k = "weight packing"
        elif "rows_" in name or "cols_" in name:
            k = "angular-spectrum passes"
        else:
            k = "other"
        fam[k] += us / 1e3
        launches[k] += n
    return fam, launches


def main():
    out = [BEGIN, "", f"*Generated from `profiles/{TAG}_*` by `tools/design_numbers.py` — edit the records, not this block.*", ""]
    b = load_json(f"{TAG}_bench_line.json")
    if b:
        r = b["roofline"]
        iso = r.get("isolated", {})
        out += [f"**Bench line** (`profiles/{TAG}_bench_line.json`; {b['config']['workload']}):", "",
                "| quantity | value |", "|---|---|",
                f"| step | **{b['ms_per_step']:.2f} ms = {b['value']:.1f} frames/s** (`dtype`: {b['dtype']}) |",
                f"| host time to enqueue a step | {b.get('host_ms_per_step', float('nan')):.1f} ms ({'one hipGraph replay' if b.get('graph') else 'eager launches'}): the step is GPU-bound |",
                f"| gather-GEMM under the timed conditions | {r['achieved']:.1f} TFLOP/s of algorithmic fp32 work = **{r['frac']:.3f}** of {r['peak']} ({r['kernel_ms_per_step']:.2f} ms/step in {r['launches_per_step']:.0f} launches, {r['algorithmic_gflop_per_step']:.0f} GFLOP) |",
                f"| gather-GEMM, second stream off | {iso.get('achieved', 0):.1f} TFLOP/s = {iso.get('frac', 0):.3f} ({iso.get('kernel_ms_per_step', 0):.2f} ms/step) |",
                f"| weight-gradient GEMM | {r['wgrad_kernel']['achieved']:.1f} TFLOP/s timed, {iso.get('wgrad_kernel', {}).get('achieved', 0):.1f} isolated ({iso.get('wgrad_kernel', {}).get('kernel_ms_per_step', 0):.2f} ms/step, {r['wgrad_kernel']['algorithmic_gflop_per_step']:.0f} GFLOP) |",
                f"| HBM traffic per gather-GEMM launch (PMC) | {(r.get('traffic') or 0) / 1e6:.1f} MB (`{(r.get('traffic_source') or {}).get('file')}`, matches this build: {(r.get('traffic_source') or {}).get('matches_this_build')}) |"]
        for name, m in (b.get("roofline_by_mode") or {}).items():
            if isinstance(m, dict):
                out.append(f"| mode `{name}`{' (headline)' if m.get('headline') else ''} | {m['ms_per_step']:.2f} ms/step; gather-GEMM {m['achieved']:.1f} TFLOP/s = {m['frac']:.3f} of {m['peak']} |")
        if b.get("secondary"):
            s = b["secondary"]
            out.append(f"| 4K (configs[3]) | {s['ms_per_frame']:.1f} ms/frame = {s['value']:.1f} frames/s; gather-GEMM {s['roofline']['achieved']:.0f} TFLOP/s = {s['roofline']['frac']:.3f} |")
        if b.get("reference_cli_default_step"):
            c = b["reference_cli_default_step"]
            out.append(f"| reference CLI's default step (d_ratio 5, perceptual 0.1; informational) | {c['ms_per_step']:.1f} ms/step = {c['value']:.1f} frames/s |")
        if b.get("cpu_baseline"):
            c = b["cpu_baseline"]
            out.append(f"| CPU baseline ({c['kind']}, {c['cores']} threads) | {c['value']:.3f} frames/s — {c['sample']} |")
        out.append("")
    head, rows = steady_table()
    if rows:
        fam, launches = family_ms(rows)
        out += [f"**Steady-state kernel profile** (`profiles/{TAG}_kernel_steady.txt`; rocprofv3 serialises the two streams) — {head}:", "",
                "| kernel family | ms/step | launches/step |", "|---|---|---|"]
        out += [f"| {k} | {v:.2f} | {launches[k]:.0f} |" for k, v in fam.items() if v > 0]
        out.append("")
    t = load_json(f"{TAG}_pmc_traffic.json")
    if t:
        g, w = t["kernels"]["gg"], t["kernels"]["wg"]
        out += [f"**HBM traffic per launch** (`profiles/{TAG}_pmc_traffic.json`, git {t.get('git_sha')}): gather-GEMM {g['hbm_bytes_per_launch_corrected'] / 1e6:.0f} MB "
                f"(read 2 x {g['fetch_kib_per_launch'] / 1024:.0f} MiB, written {g['write_kib_per_launch'] / 1024:.0f} MiB); weight-gradient GEMM "
                f"{w['hbm_bytes_per_launch_corrected'] / 1e6:.0f} MB of which **{w['write_kib_per_launch'] / 1024:.1f} MiB written** (the split-K slabs).", ""]
    path = os.path.join(REPO, "profiles", f"{TAG}_wg6_sweep.jsonl")
    if os.path.exists(path):
        out += [f"**Tap-fused weight-gradient GEMM per layer** (`profiles/{TAG}_wg6_sweep.jsonl`, `tools/wg6_sweep.py`: GEMM + reduction, isolated, random operands; "
                f"per-tap kernels: `profiles/{TAG}_wg6_sweep_legacy.txt`):", "", "| layer | best variant / splits | us | TFLOP/s | launches per step |", "|---|---|---|---|---|"]
        tot = 0.0
        for ln in open(path):
            if not ln.strip():
                continue
            r_ = json.loads(ln)
            vs = [v for v in r_["variants"] if not (v["S"] > 1 and v["fused"])] or r_["variants"]
            if not vs:
                continue
            best = min(vs, key=lambda v: v["us"])
            tot += best["us"] * r_["count"]
            out.append(f"| {r_['layer']} | {best['name']}, S = {best['S']} | {best['us']:.0f} | {r_['flops'] / best['us'] / 1e6:.0f} | {r_['count']} |")
        out += ["", f"Sum over the step's launches at the best variant: {tot / 1e3:.2f} ms.", ""]
    path = os.path.join(REPO, "profiles", f"{TAG}_4k_kernel_steady.txt")
    if os.path.exists(path):
        lines4 = open(path).read().splitlines()
        out += [f"**4K frame** (`profiles/{TAG}_4k_kernel_steady.txt`, `{TAG}_4k_pmc_traffic.json`) — {lines4[0]}:", "", "| kernel family | launches/frame | ms/frame | share |", "|---|---|---|---|"]
        for ln in lines4[1:]:
            m = re.match(r"\s+family (.+?)\s+([\d.]+)/frame\s+([\d.]+) ms/frame\s+([\d.]+)%", ln)
            if m:
                out.append(f"| {m.group(1).strip()} | {float(m.group(2)):.0f} | {float(m.group(3)):.2f} | {m.group(4)} % |")
        out.append("")
    path = os.path.join(REPO, "profiles", f"{TAG}_truth_tests.jsonl")
    if os.path.exists(path):
        recs = [json.loads(ln) for ln in open(path) if ln.strip()]
        full = [r_ for r_ in recs if r_["test"] == "full_size_step"]
        if full:
            keys = ("hat_amps_max", "hat_amps_l2", "poh_q999", "G_loss", "D_loss")
            out += [f"**Distance to the float64 evaluation, full-size step** (`profiles/{TAG}_truth_tests.jsonl`; GPU / CPU-fp32, ratio):", "",
                    "| mode | " + " | ".join(keys) + " |", "|---|" + "---|" * len(keys)]
            for r_ in full:
                out.append(f"| `{r_['mode']}` | " + " | ".join(f"{r_[k][0]:.1e} / {r_[k][1]:.1e} ({r_[k][0] / max(r_[k][1], 1e-300):.2f})" for k in keys) + " |")
            out.append("")
            if any("grad_G_l2" in r_ for r_ in full):  # round 5: every parameter's gradient at the bench size
                gkeys = ("grad_G_l2", "grad_G_worst_parameter", "grad_D_l2", "grad_D_worst_parameter")
                out += [f"**Parameter gradients at the bench size against float64** (same records; relative L2, GPU / CPU-fp32 (ratio); `*_l2`: all parameters of a model as "
                        "one vector, `*_worst_parameter`: the parameter (or the group of parameters with < 16 elements) with the largest ratio):", "",
                        "| mode | " + " | ".join(gkeys) + " | three worst by name (ratio) |", "|---|" + "---|" * (len(gkeys) + 1)]
                for r_ in full:
                    if "grad_G_l2" not in r_:
                        continue
                    names = "; ".join(f"{t_}: {w_['name']} ({w_['ratio']:.1f})" for t_ in ("G", "D") for w_ in r_.get(f"worst_parameters_{t_}", [])[:3])
                    out.append(f"| `{r_['mode']}` | " + " | ".join(f"{r_[k][0]:.1e} / {r_[k][1]:.1e} ({r_[k][0] / max(r_[k][1], 1e-300):.2f})" for k in gkeys) + f" | {names} |")
                out.append("")
    path = os.path.join(REPO, "profiles", f"{TAG}_pmc_mfma.jsonl")
    if os.path.exists(path):
        out += [f"**Matrix-pipe occupancy** (`profiles/{TAG}_pmc_mfma.jsonl`):", "", "| layer (tag: mode_Cin_Cout_extent) | kernel matched | us | matrix pipe busy | clock GHz |", "|---|---|---|---|---|"]
        for ln in open(path):
            if ln.strip().startswith("{"):
                d = json.loads(ln)
                out.append(f"| {d['tag']} | {d['kernel']} | {d['us_under_pmc'][-1]} | {d.get('matrix_pipe_busy')} | {d.get('shader_clock_ghz')} |")
        out.append("")
    out.append(END)
    path = os.path.join(REPO, "DESIGN.md")
    text = open(path).read()
    block = "\n".join(out)
    if BEGIN in text and END in text:
        text = text[:text.index(BEGIN)] + block + text[text.index(END) + len(END):]
    else:
        text = text.rstrip("\n") + "\n\n" + block + "\n"
    open(path, "w").write(text)
    print(block)


if __name__ == "__main__":
    main()
