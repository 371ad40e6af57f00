export AGZ_WINO_VARIANT=6
(timeout 600 python -m pytest tests -m gpu -x -q -k "gpu_nn.py" 2>&1 | grep -E "passed|failed" | tail -2)
python tools/nn_micro.py --batches 8192 --algos 1 --iters 10 2>&1 | grep forward_ms | cut -c1-100
for x in 1; do echo -n "X=$x "; AGZ_WINO_X=$x python tools/nn_micro.py --batches 8192 --algos 1 --iters 5 2>&1 | grep forward_ms | cut -c1-100; done
AGZ_WINO_TRACE=gpurun_out/g6_t_two.bin python tools/nn_micro.py --batches 8192 --algos 1 --iters 2 2>&1 | tail -1 | cut -c1-60
AGZ_WINO_ONE=1 AGZ_WINO_TRACE=gpurun_out/g6_t_one.bin python tools/nn_micro.py --batches 8192 --algos 1 --iters 2 2>&1 | tail -1 | cut -c1-60
unset AGZ_WINO_VARIANT
python tools/nn_micro.py --batches 8192 --algos 1 --iters 10 2>&1 | grep forward_ms | cut -c1-100
