export TMPDIR=/tmp; cd /tmp
R=$GRAFT_REPO_ROOT
for L in spec1 spec6; do
  for W in walk rand; do
    if [ $W = walk ]; then T="$R/scripts/gpu_walk_target.py f32"; else T="$R/scripts/gpu_pmc_target.py 50 4096 f32"; fi
    PLENVEC_LIB=$R/plen_ml_walk_amd/csrc/variants/$L.so timeout 300 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE --output-format csv -d $R/gpurun_out/icache_${L}_$W -- python3 $T > $R/gpurun_out/icache_${L}_$W.log 2>&1
  done
done
