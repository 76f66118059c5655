set -x
mkdir -p gpurun_out/r02a
python -m pytest tests/test_multi_device.py tests/test_full_size.py::test_long_reads_configs3_shape_round_trip_and_sampled_zlib_parity tests/test_gpu_parity.py -x -q 2>&1 | tail -15 > gpurun_out/r02a/pytest.txt
cat gpurun_out/r02a/pytest.txt
python bench.py > gpurun_out/r02a/bench_default.json 2> gpurun_out/r02a/bench_default.err; tail -c 3000 gpurun_out/r02a/bench_default.json; tail -5 gpurun_out/r02a/bench_default.err
tools/kstats.sh r02a/svb python bench.py --svb-only --cpu-seconds 0
tools/kstats.sh r02a/long python bench.py --long --cpu-seconds 0
tools/kstats.sh r02a/mixed python bench.py --mixed --cpu-seconds 0
tools/kstats.sh r02a/decode python bench.py --decode
tools/kstats.sh r02a/zstd python tools/zstd_time.py 1000000 4000
tools/kstats.sh r02a/enc python bench.py --no-long --cpu-seconds 0
