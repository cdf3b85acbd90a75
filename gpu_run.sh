for pr in 0 1 3; do echo "== stagger $pr"; MK_PROBE=$pr timeout 120 python tools/microbench.py dhconv 2>&1 | grep -E "dhconv" | cut -c1-60; done
