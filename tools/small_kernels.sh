# the small configurations after a kernel / dispatch change: parity tests of the 2-D models, then the long rollouts
OUT=${1:-gpurun_out/r05s}; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_model.py -q -x -k "config or rollout" 2>&1 | tail -n 2
for r in "waterramps 600" "wbcsph 3200"; do set -- $r; timeout 900 python tools/long_rollout.py $1 $2 --out $OUT/rollout_$1.json >> $OUT/rollouts.log 2>&1; tail -n 1 $OUT/rollouts.log | cut -c1-600; done
