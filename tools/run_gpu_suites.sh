# The GPU suite under the exact and the split policy with measured-cap recording (tools/update_caps.py reads the two metrics files), then smoke()
cd $GRAFT_REPO_ROOT
O=gpurun_out/leg; mkdir -p $O
rm -f gpurun_out/test_metrics.jsonl
ROBIR_RECORD_CAPS=1 python -m pytest tests -m gpu -q > $O/suite_exact.log 2>&1; tail -2 $O/suite_exact.log; grep "^FAILED\|^ERROR" $O/suite_exact.log | head -20
cp gpurun_out/test_metrics.jsonl $O/metrics_exact.jsonl; rm -f gpurun_out/test_metrics.jsonl
ROBIR_RECORD_CAPS=1 ROBIR_PRECISION=split python -m pytest tests -m gpu -q > $O/suite_split.log 2>&1; tail -2 $O/suite_split.log; grep "^FAILED\|^ERROR" $O/suite_split.log | head -20
cp gpurun_out/test_metrics.jsonl $O/metrics_split.jsonl; rm -f gpurun_out/test_metrics.jsonl
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
