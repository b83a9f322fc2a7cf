#!/bin/bash
# round 6, GPU call C: the 12-slot Y buffer + packed factor (2 and 3 waves per SIMD) against the round-5 build
set -u
OUT=gpurun_out/r06_${1:-c}
mkdir -p $OUT
timeout 900 python scripts/gpu_same_bits.py r05.so 2>&1 | grep -v amdgpu.ids > $OUT/same_bits.txt
cat $OUT/same_bits.txt
PLENVEC_LIB=$PWD/plen_ml_walk_amd/csrc/variants/wpe3.so timeout 900 python scripts/gpu_same_bits.py r05.so 2>&1 | grep -v amdgpu.ids > $OUT/same_bits_wpe3.txt
cat $OUT/same_bits_wpe3.txt
timeout 900 python scripts/gpu_ab64.py r05.so - wpe3.so > $OUT/ab_f64.txt 2>&1
cat $OUT/ab_f64.txt
timeout 900 python -m pytest tests/test_env_gpu.py tests/test_full_size_gpu.py tests/test_box_contacts_gpu.py tests/test_cabi_gpu.py -m gpu -x -q > $OUT/gputest_env.txt 2>&1; echo "pytest rc $?" >> $OUT/gputest_env.txt
tail -5 $OUT/gputest_env.txt
AB_DTYPE=f32 timeout 900 python scripts/gpu_ab64.py r05.so - > $OUT/ab_f32.txt 2>&1
cat $OUT/ab_f32.txt
