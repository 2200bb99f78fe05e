mkdir -p gpurun_out/q
timeout -k 10 600 python -m pytest tests/test_gpu_capi.py tests/test_gpu_backend.py tests/test_gpu_configs_full_size.py tests/test_gpu_fuzz.py -m gpu -x -q > gpurun_out/q/tests.log 2>&1 || { tail -20 gpurun_out/q/tests.log; exit 1; }
python tools/bench_quotient_phases.py > gpurun_out/q/phases.log 2>&1
TSTWO_QUOT_NO_LAZY=1 python tools/bench_quotient_phases.py > gpurun_out/q/phases_nolazy.log 2>&1
tail -1 gpurun_out/q/tests.log; tail -1 gpurun_out/q/phases.log; tail -1 gpurun_out/q/phases_nolazy.log
