timeout 600 python -m pytest tests/test_bench_contract.py -m gpu -q --tb=short 2>&1 | tail -3
