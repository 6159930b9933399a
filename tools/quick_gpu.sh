#!/bin/bash
# usage: tools/quick_gpu.sh <pytest -k expr or file>  -- convenience wrapper around gpurun for the parity suite
/usr/local/graft/bin/gpurun --timeout ${TIMEOUT:-900} -- "python -m pytest $* -m gpu -q 2>&1 | grep -E '^E |Error|passed|failed|^FAILED' | head -60"
