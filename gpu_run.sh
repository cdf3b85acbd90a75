timeout 300 python -m pytest tests/test_losses.py -m gpu -q --tb=short -x 2>&1 | tail -3
