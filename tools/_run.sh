R=${GRAFT_REPO_ROOT:-/root/repo}
timeout 900 python -m pytest $R/tests -m gpu -x -q 2>&1 | tail -2
echo "dolly: $(DOLLY=1 timeout 120 python $R/tools/throughput.py 2>&1 | grep throughput)"
echo "raw: $(CLOUD=raw timeout 120 python $R/tools/throughput.py 2>&1 | grep throughput)"
echo "default: $(timeout 120 python $R/tools/throughput.py 2>&1 | grep throughput)"
