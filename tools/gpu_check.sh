# usage (on the GPU box): bash tools/gpu_check.sh  -- GPU test suite, then a short headline bench summary
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/gpu_tests.log 2>&1; echo "pytest exit=$?"; tail -2 gpurun_out/gpu_tests.log
python bench.py --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['stage_ms'])"
