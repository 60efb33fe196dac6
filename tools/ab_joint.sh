#!/bin/bash
# A/B of library variants in one GPU-box call on the whole-model training step (bench.py --scope joint, captured).
for rep in 1 2; do
for tag in "$@"; do
    CGVP_LIB_PATH=$PWD/caster-dta_amd/lib/ab/libcaster_gvp_$tag.so python bench.py --scope joint --no-cpu-baseline --epoch off --steps 100 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); print('%-10s joint ms_per_step %.4f' % ('$tag', d['ms_per_step']))"
done
done
