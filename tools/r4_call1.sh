export TMPDIR=/tmp; mkdir -p gpurun_out
tools/catchup.sh quick 2>&1 | tee gpurun_out/c1_quick.txt
timeout 1500 python -m pytest tests -m gpu -q --timeout 600 --timeout-method=thread -p no:cacheprovider 2>&1 | tail -25 > gpurun_out/c1_tests_verified.txt
SNK_RUN_UNVERIFIED=1 timeout 900 python -m pytest tests/test_adapter_fuzz_gpu.py tests/test_rmdup_gpu.py tests/test_gunzip_gpu.py tests/test_cli_gpu.py -m gpu -q --timeout 300 --timeout-method=thread -p no:cacheprovider 2>&1 | tail -40 > gpurun_out/c1_tests_unverified.txt
tail -5 gpurun_out/c1_tests_verified.txt; tail -30 gpurun_out/c1_tests_unverified.txt
