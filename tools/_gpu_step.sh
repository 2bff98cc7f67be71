set -e -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "conv" > gpurun_out/s2_conv_tests.log 2>&1 || { tail -40 gpurun_out/s2_conv_tests.log; exit 1; }
tail -3 gpurun_out/s2_conv_tests.log
timeout -k 10 400 python -m pytest tests/test_gpu_model.py tests/test_gpu_pipeline.py -x -q -m gpu > gpurun_out/s2_model_tests.log 2>&1 || { tail -30 gpurun_out/s2_model_tests.log; exit 1; }
tail -3 gpurun_out/s2_model_tests.log
python tools/bench_head.py 10 | tee gpurun_out/s2_head_c.log
export TMPDIR=/tmp
R=$PWD
rm -rf gpurun_out/prof_head
cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_head -o head -- python3 $R/tools/bench_head.py 5 > $R/gpurun_out/prof_head.log 2>&1
cd $R && python tools/bench_head.py --summarize gpurun_out/prof_head | tee gpurun_out/s2_head_launches_c.txt
