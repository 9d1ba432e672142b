timeout 1500 python -m pytest tests/test_gpu_fullsize.py -q -m gpu -s -k "contract" 2>&1 | grep -E "^conditioned|passed|failed|Error"
