python -m pytest tests/test_bench_contract.py -m gpu -x -q 2>&1 | tail -15
