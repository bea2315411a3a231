# the small configurations after a kernel / dispatch change: parity tests of the 2-D models, then the three long rollouts
OUT=${1:-gpurun_out/r05s}; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_ops.py -q -x -k "one_chunk_per_workgroup" 2>&1 | tail -n 2
timeout 1500 python -m pytest tests/test_gpu_model.py -q -x -k "config or rollout" 2>&1 | tail -n 3
for r in "liquid3d_dam 200" "waterramps 600" "wbcsph 3200"; do set -- $r; timeout 900 python tools/long_rollout.py $1 $2 --out $OUT/rollout_$1.json >> $OUT/rollouts.log 2>&1; tail -n 2 $OUT/rollouts.log; done
