#!/bin/bash
# strict gausslq over every box: bit-identity tests, then timings
export TMPDIR=/tmp
timeout 1500 python3 -m pytest tests/test_gpu_parity.py -x -q -k "gausslq" 2>&1 | tail -6
for b in ${BOXES:-3 5 7 9 11 13 15 17 21}; do python3 tools/time_gausslq.py ${NSPOTS:-1048576} $b 2>&1 | grep -v amdgpu.ids | tail -4 | sed -n '2p;4p' | sed "s/^/[box $b] /"; done
