set -e -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_unet.py -x -q -m gpu -s > gpurun_out/s2_unet_tests.log 2>&1 || { tail -60 gpurun_out/s2_unet_tests.log; exit 1; }
tail -8 gpurun_out/s2_unet_tests.log
