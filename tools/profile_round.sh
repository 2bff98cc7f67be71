#!/bin/bash
# Round profiles, run ON THE GPU BOX (gpurun): kernel-trace stats of the benchmark command, then the PMC passes of the dominant
# kernel (the SwiGLU w12 GEMM alone, tools/run_one_gemm.py) -- counters in their own runs, never combined with tracing.
#   bash tools/profile_round.sh r02        -> gpurun_out/prof_r02/...   (then, in the BUILD container where gpurun merged the files back:
#   python tools/summarize_profiles.py r02 copies the summaries into the tracked profiles/ directory)
set -e -o pipefail
TAG=${1:-r02}
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
REPO=$PWD
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o bench -- python3 "$REPO/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --no-stages > "$OUT/bench_under_rocprof.log" 2> "$OUT/bench_under_rocprof.err"
echo "[profile] kernel stats done"
export CVX_ABLATION_LIB=0
for pass in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE" "SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE"; do
    name=$(echo "$pass" | cut -d' ' -f1)
    rocprofv3 --pmc $pass --output-format csv -d "$OUT/pmc_$name" -o pmc -- python3 "$REPO/tools/run_one_gemm.py" > "$OUT/pmc_$name.log" 2>&1
    echo "[profile] pmc $name done"
done
# attention: instruction mix and pipe occupancy
for pass in "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVES GRBM_GUI_ACTIVE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE"; do
    name=attn_$(echo "$pass" | cut -d' ' -f1)
    rocprofv3 --pmc $pass --output-format csv -d "$OUT/pmc_$name" -o pmc -- python3 "$REPO/tools/run_one_attention.py" > "$OUT/pmc_$name.log" 2>&1
    echo "[profile] pmc $name done"
done
# configs[2] alone: per-launch table of the head
rocprofv3 --kernel-trace --output-format csv -d "$OUT/head" -o head -- python3 "$REPO/tools/bench_head.py" 5 > "$OUT/head.log" 2>&1
echo "[profile] head trace done"
cd "$REPO"
python3 tools/bench_head.py --summarize "$OUT/head" > "$OUT/head_launches.txt"
python3 tools/summarize_profiles.py "$TAG"
