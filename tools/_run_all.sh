cd /root/repo
timeout -k 10 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 > gpurun_out/r04_pytest_gpu.log; cat gpurun_out/r04_pytest_gpu.log
timeout -k 10 500 python bench.py > gpurun_out/r04_bench_default.json 2> gpurun_out/r04_bench_default.err; echo "default rc=$?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04_bench_default.json').read().strip().splitlines()[-1])
print(d['value'], d['fit']['cholesky_s'], d['fit']['cholesky_GFLOPs'], d['fit']['variance_prep_s'], d['fit']['trtri_s'], d['peak_hbm_bytes_per_rank'], d['parity'].get('vs_sklearn_at_n_train'))
PY
timeout -k 10 900 python tools/run_configs.py > gpurun_out/r04_baseline_configs.log 2>&1; echo "configs rc=$?"; grep "^C[0-9]" gpurun_out/r04_baseline_configs.log | cut -c1-900
