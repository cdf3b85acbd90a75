timeout 600 python -m pytest tests -m gpu -q --tb=short -k "golden_fp32" 2>&1 | tail -25
