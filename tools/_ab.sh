python -m pytest tests/test_gpu_parity.py -q -m gpu -k "gemv or fused or moe or expert or linear or decoder" 2>&1 | grep -E "^FAILED|^E  |passed|failed" | head -40
