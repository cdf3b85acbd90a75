timeout 900 python -m pytest tests -m gpu -q --tb=short 2>&1 | tail -30
