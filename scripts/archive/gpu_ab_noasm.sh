for r in 1 2; do
for v in 0 1; do
  if [ $v = 1 ]; then export PLENVEC_NO_ASM=1; else unset PLENVEC_NO_ASM; fi
  python bench.py --dtype f64 --legs "" --no-cpu-baseline --no-parity --steps 200 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('NO_ASM=$v pipelined %.3f M (%.4f ms) one-launch %.4f ms kernel %.4f'%(d['value']/1e6,d['ms_per_step'],d['config']['one_launch_per_step']['ms_per_step'],d['kernel_ms_per_launch']))"
done; done
unset PLENVEC_NO_ASM
python -m pytest tests/test_env_gpu.py -x -q -m gpu -k "asm or solver or rollout" 2>&1 | tail -3
