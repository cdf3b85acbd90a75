timeout 900 python -m pytest tests -m gpu -q --tb=short -x 2>&1 | tail -3
