rm -f gpurun_out/test_metrics.jsonl
ROBIR_RECORD_CAPS=1 timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -3
ROBIR_RECORD_CAPS=1 ROBIR_PRECISION=split timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -3
wc -l gpurun_out/test_metrics.jsonl
