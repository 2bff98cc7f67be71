"""Condense gpurun_out/prof_<tag>/ (tools/profile_round.sh) into the tracked files under profiles/:
  profiles/<tag>_kernel_stats.csv          rocprofv3 --kernel-trace --stats of `bench.py --steps 3 --warmup 1`, top kernels
  profiles/<tag>_pmc_dominant.csv          the dominant kernel's rows of the four --pmc passes (per dispatch)
  profiles/dominant_kernel_traffic.json    per-launch means the bench line quotes (FETCH_SIZE doubled: gfx950 correction)
"""
import csv
import glob
import json
import sys
from pathlib import Path

csv.field_size_limit(1 << 30)
ROOT = Path(__file__).resolve().parent.parent
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
src = ROOT / "gpurun_out" / f"prof_{tag}"
dst = ROOT / "profiles"
DOM = "k_gemm256p_nreg<cvx::EpiSwiGLU"  # (EpiSwiGLUT<true>: the LN-folded form the ViT path launches)

stats = glob.glob(str(src / "stats" / "**" / "*kernel_stats.csv"), recursive=True)
if stats:
    rows = list(csv.DictReader(open(stats[0])))
    rows = [r for r in rows if float(r["Percentage"]) >= 0.05][:40]
    with open(dst / f"{tag}_kernel_stats.csv", "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=list(rows[0]))
        w.writeheader()
        w.writerows(rows)
    for r in rows[:12]:
        print(f"{r['Name'][:90]:90s} {int(r['Calls']):5d} avg {float(r['AverageNs']) / 1e3:9.1f} us {float(r['Percentage']):5.1f}%")
for name in ("bench_under_rocprof.log",):
    if (src / name).exists():
        (dst / f"{tag}_{name}").write_text((src / name).read_text())

if (src / "head_launches.txt").exists():
    (dst / f"{tag}_head_launches.txt").write_text((src / "head_launches.txt").read_text())

# ---- attention kernel: instruction mix / pipe occupancy per launch ----
avals: dict[str, list[float]] = {}
arows = []
for f in glob.glob(str(src / "pmc_attn_*" / "**" / "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_attention" in r["Kernel_Name"]:
            avals.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
            arows.append({"dispatch": r["Dispatch_Id"], "counter": r["Counter_Name"], "value": r["Counter_Value"], "start_ns": r["Start_Timestamp"],
                          "end_ns": r["End_Timestamp"], "vgpr": r["VGPR_Count"], "lds": r["LDS_Block_Size"], "grid": r["Grid_Size"], "wg": r["Workgroup_Size"]})
if arows:
    with open(dst / f"{tag}_pmc_attention.csv", "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=list(arows[0]))
        w.writeheader()
        w.writerows(arows)
    am = {k: sum(v[1:]) / max(1, len(v) - 1) for k, v in avals.items()}
    a = {"kernel": "k_attention (default variant: cvx_set_option attn_variant), 128 slices x 24 heads x 1029 tokens, head_dim 64", "per_launch_means": am}
    if "SQ_INSTS_MFMA" in am and "SQ_INSTS_VALU" in am:
        a["valu_per_mfma"] = am["SQ_INSTS_VALU"] / am["SQ_INSTS_MFMA"]
    if "SQ_VALU_MFMA_BUSY_CYCLES" in am and "SQ_BUSY_CYCLES" in am:
        a["note"] = "SQ_VALU_MFMA_BUSY_CYCLES / SQ_ACTIVE_INST_VALU are summed over SIMDs; compare with 4 x SQ_BUSY_CYCLES-per-SE totals as in DESIGN.md s.4"
    (dst / f"{tag}_attention_counters.json").write_text(json.dumps(a, indent=1))
    print(json.dumps(a, indent=1))

vals: dict[str, list[float]] = {}
dur: list[float] = []
out_rows = []
for f in glob.glob(str(src / "pmc_*" / "**" / "*counter_collection.csv"), recursive=True):
    if "pmc_attn_" in f:
        continue
    for r in csv.DictReader(open(f)):
        if DOM in r["Kernel_Name"]:
            vals.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
            out_rows.append({"pass": next(q for q in reversed(Path(f).parts[:-1]) if q.startswith("pmc_")), "dispatch": r["Dispatch_Id"],
                             "counter": r["Counter_Name"], "value": r["Counter_Value"], "start_ns": r["Start_Timestamp"], "end_ns": r["End_Timestamp"],
                             "vgpr": r["VGPR_Count"], "lds": r["LDS_Block_Size"], "grid": r["Grid_Size"], "wg": r["Workgroup_Size"]})
            if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
                dur.append((float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e6)
if out_rows:
    with open(dst / f"{tag}_pmc_dominant.csv", "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=list(out_rows[0]))
        w.writeheader()
        w.writerows(out_rows)
    mean = lambda k: sum(vals[k][1:]) / max(1, len(vals[k]) - 1)  # noqa: E731  (the first launch is the cold one)
    M, K, N = 128 * 1032, 1536, 8192
    alg = M * K * 2 + N * K * 2 + M * (N // 2) * 2 + M * 8 + 2 * N * 4  # + row constants and b' / column sums of the fold
    fetch, write = mean("FETCH_SIZE") * 1024 * 2, mean("WRITE_SIZE") * 1024
    ms = sum(dur[1:]) / max(1, len(dur) - 1)
    import subprocess

    try:
        commit = subprocess.run(["git", "rev-parse", "--short", "HEAD"], cwd=ROOT, capture_output=True, text=True).stdout.strip()
    except OSError:
        commit = ""
    d = {
        "kernel": "k_gemm256p_nreg<EpiSwiGLUT<LN = true>, FULL> (persistent 256x256x64 tile, gemm256p.h; LayerNorm folded into the epilogue)",
        "measured_at_commit": commit,  # HEAD of the tree whose gpurun_out/prof_<tag> was summarised (the kernel sources it was built from)
        "tag": tag,
        "shape": f"M={M} (128 slices x 1032 padded tokens), K={K}, N={N} -> out bf16 [M,{N // 2}]",
        "collection": f"rocprofv3 --pmc, separate passes (FETCH_SIZE | WRITE_SIZE | TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE | SQ_*), "
                      f"tools/run_one_gemm.py on the product library, mean of launches 2..6; rows: profiles/{tag}_pmc_dominant.csv",
        "fetch_bytes_per_launch": fetch, "write_bytes_per_launch": write, "algorithmic_bytes_per_launch": alg,
        "note": "FETCH_SIZE (KB) doubled per the guide's gfx950 correction for wide coalesced streams; it is a fabric-side (L2 miss) counter "
                "that INCLUDES Infinity-Cache hits, i.e. an upper bound on HBM reads: every XCD pulls the activation panels of its 8x4 tile "
                "block once per round through its private L2 (~4.9 GB by construction); WRITE_SIZE = the output bytes",
        "l2_hit_rate": mean("TCC_HIT_sum") / (mean("TCC_HIT_sum") + mean("TCC_MISS_sum")),
        "effective_clock_ghz_under_profiling": mean("GRBM_GUI_ACTIVE") / 8 / (ms * 1e-3) / 1e9,
        # SQ_VALU_MFMA_BUSY_CYCLES: cycles, summed over the 1024 SIMDs (= 16 x the number of 16x16x32 MFMAs);
        # SQ_WAVE_CYCLES: quad-cycles summed over the 2048 resident waves (256 workgroups x 8) -> busy / (1024 x kernel cycles)
        "mfma_busy_fraction": mean("SQ_VALU_MFMA_BUSY_CYCLES") / (2.0 * mean("SQ_WAVE_CYCLES")) if "SQ_WAVE_CYCLES" in vals else None,
        "wave_wait_fraction": mean("SQ_WAIT_ANY") / mean("SQ_WAVE_CYCLES") if "SQ_WAIT_ANY" in vals else None,
        "wave_issue_stall_fraction": mean("SQ_WAIT_INST_ANY") / mean("SQ_WAVE_CYCLES") if "SQ_WAIT_INST_ANY" in vals else None,
        "lds_bank_conflict_cycles_per_launch": mean("SQ_LDS_BANK_CONFLICT") if "SQ_LDS_BANK_CONFLICT" in vals else None,
        "avg_launch_ms_under_pmc": ms,
    }
    (dst / "dominant_kernel_traffic.json").write_text(json.dumps(d, indent=1))
    print(json.dumps(d, indent=1))
