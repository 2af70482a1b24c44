cd $GRAFT_REPO_ROOT
echo "== overflow default"; timeout 300 python -m pytest tests/test_gpu_parity.py -q -x -k "overflow" 2>&1 | tail -3
echo "== overflow chain=0"; DSBDD_NODE_CHAIN=0 timeout 300 python -m pytest tests/test_gpu_parity.py -q -x -k "overflow" 2>&1 | tail -3
echo "== chain test"; timeout 600 python -m pytest tests/test_gpu_fullsize.py -q -k "node_chain" 2>&1 | tail -15
