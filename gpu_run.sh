timeout 300 python -m pytest tests/test_gpu_model.py -m gpu -q --tb=short -x -k "shadow" 2>&1 | tail -3
