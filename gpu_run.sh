mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_fcn3.py -m gpu -q -s -k "whole_network or local_block" 2>&1 | grep -v "^$" | tail -12
