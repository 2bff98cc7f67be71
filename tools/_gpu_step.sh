set -e -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_train.py -x -q -m gpu > gpurun_out/s2_train_tests.log 2>&1 || { tail -40 gpurun_out/s2_train_tests.log; exit 1; }
tail -3 gpurun_out/s2_train_tests.log
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/s2_all_tests.log 2>&1 || { tail -40 gpurun_out/s2_all_tests.log; exit 1; }
tail -3 gpurun_out/s2_all_tests.log
python bench.py --steps 4 --warmup 1 > gpurun_out/s2_bench_a.json 2> gpurun_out/s2_bench_a.err || { tail -20 gpurun_out/s2_bench_a.err; exit 1; }
cat gpurun_out/s2_bench_a.json
