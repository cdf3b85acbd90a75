timeout 600 python -m pytest tests -m gpu -q --tb=short -k "s2norm or s2_norm" 2>&1 | tail -25
