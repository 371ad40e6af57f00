#!/bin/bash
# every tracked round-4 profile in one go (GPU box, repo root): ~20 minutes.  Then tools/collect_r04_profiles.sh here.
set -u
mkdir -p gpurun_out/r04lines
python -m pytest tests -m gpu -x -q 2>&1 | tail -30 > gpurun_out/r04lines/gpu_tests.log
tools/profile_r04.sh f32 1024 9 > gpurun_out/r04prof_f32.log 2>&1
tools/profile_r04.sh c4 256 19 --board 19 --tower 20 --readouts 800 --games 256 > gpurun_out/r04prof_c4.log 2>&1
tools/profile_r04.sh c5 512 19 --board 19 --tower 20 --readouts 1600 --games 512 --precision f16 > gpurun_out/r04prof_c5.log 2>&1
tools/profile_r04.sh f16 1024 9 --precision f16 > gpurun_out/r04prof_f16.log 2>&1
tools/profile_r04.sh f32s 1024 9 --precision f32s > gpurun_out/r04prof_f32s.log 2>&1
python bench.py --steps 100 --warmup 10 > gpurun_out/r04lines/bench_f32.json 2> gpurun_out/r04lines/bench_f32.err
python bench.py --board 19 --tower 20 --readouts 800 --games 256 --steps 40 --warmup 5 --no-cpu-baseline > gpurun_out/r04lines/c4_bench.json 2>/dev/null
python bench.py --board 19 --tower 20 --readouts 800 --games 256 --steps 40 --warmup 5 --no-cpu-baseline --no-alt-precision --winograd 2 > gpurun_out/r04lines/c4_bench_f33.json 2>/dev/null
python bench.py --board 19 --tower 20 --readouts 1600 --games 512 --precision f16 --steps 40 --warmup 5 --no-cpu-baseline > gpurun_out/r04lines/c5_bench.json 2>/dev/null
python bench.py --precision f16 --steps 60 --warmup 5 --no-cpu-baseline > gpurun_out/r04lines/bench_f16.json 2>/dev/null
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --single-device-test --steps 20 --warmup 3 > gpurun_out/r04lines/bench_2rank_single_device.json 2> gpurun_out/r04lines/bench_2rank.err
python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 8 --single-device-test --games 128 --steps 10 --warmup 2 > gpurun_out/r04lines/bench_8rank_single_device.json 2> gpurun_out/r04lines/bench_8rank.err
python tools/generation.py --out gpurun_out/r04lines/generation.json > /dev/null 2> gpurun_out/r04lines/generation.err
for f in gpurun_out/r04lines/*.json; do echo "$f: $(tail -1 $f | cut -c1-160)"; done
