set -o pipefail
O=gpurun_out/sched; mkdir -p $O
timeout -k 10 300 python -u -m pytest tests/test_scheduler_gpu.py tests/test_zx_sampling_job_gpu.py -q -x --timeout 300 --timeout-method thread > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -2 $O/tests.log
bash tools/hbm_rows_pmc.sh $O/hbm
