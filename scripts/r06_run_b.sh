#!/bin/bash
# round 6, GPU call B: a kernel build against the round-5 build bit for bit, the env tests, A/B on both headline legs.
set -u
OUT=gpurun_out/r06_${1:-b}
PREV=${2:-r05.so}
mkdir -p $OUT
timeout 600 python scripts/gpu_same_bits.py r05.so 2>&1 | grep -v amdgpu.ids > $OUT/same_bits.txt
timeout 900 python -m pytest tests/test_env_gpu.py tests/test_full_size_gpu.py tests/test_box_contacts_gpu.py tests/test_cabi_gpu.py -m gpu -x -q > $OUT/gputest_env.txt 2>&1; echo "pytest rc $?" >> $OUT/gputest_env.txt
timeout 900 python scripts/gpu_ab64.py $PREV - > $OUT/ab_f64.txt 2>&1
AB_DTYPE=f32 timeout 900 python scripts/gpu_ab64.py $PREV - > $OUT/ab_f32.txt 2>&1
cat $OUT/same_bits.txt; tail -3 $OUT/gputest_env.txt; cat $OUT/ab_f64.txt $OUT/ab_f32.txt
