timeout 22 python -m pytest tests/test_bench_contract.py -m gpu -q --tb=short -x 2>&1 | tail -6
