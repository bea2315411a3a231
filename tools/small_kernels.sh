# the small configurations after a kernel / dispatch change: parity tests of the models, then the long rollouts
OUT=${1:-gpurun_out/r05s}; mkdir -p $OUT
timeout 2400 python -m pytest tests/test_gpu_model.py -q -x 2>&1 | tail -n 2
for r in "liquid3d_dam 200" "waterramps 600" "wbcsph 3200"; do set -- $r; timeout 900 python tools/long_rollout.py $1 $2 --out $OUT/rollout_$1.json >> $OUT/rollouts.log 2>&1; echo -n "$1: "; grep -o '"repeated_steps": [0-9]*\|"ms_median": [0-9.]*' $OUT/rollout_$1.json | tr '\n' ' '; echo; done
python bench.py --cpu-side 0 2>/dev/null | tail -n 1 | cut -c1-260
