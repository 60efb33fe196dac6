#!/bin/bash
# Run the GINE parity tests against a library variant: bash tools/ab_tests.sh <tag>
CGVP_LIB_PATH=$PWD/caster-dta_amd/lib/ab/libcaster_gvp_$1.so timeout -k 10 400 python -m pytest tests/test_hip_parity.py tests/test_gine_depth4.py tests/test_hip_backward.py tests/test_hip_random_graphs.py -m gpu -q -x 2>&1 | tail -2
