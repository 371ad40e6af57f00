#!/bin/bash
# every tracked round-6 profile in one go (GPU box, repo root): ~30 minutes.  Then tools/collect_r06_profiles.sh here.
set -u
mkdir -p gpurun_out/r06lines
if [ "${SKIP_TESTS:-0}" != "1" ]; then python -m pytest tests -m gpu -x -q 2>&1 | tail -30 > gpurun_out/r06lines/gpu_tests.log; fi
tools/profile_r06.sh f32 1024 9 > gpurun_out/r06prof_f32.log 2>&1
tools/profile_r06.sh c4 256 19 --board 19 --tower 20 --readouts 800 --games 256 > gpurun_out/r06prof_c4.log 2>&1
tools/profile_r06.sh c5 512 19 --board 19 --tower 20 --readouts 1600 --games 512 --precision f16 > gpurun_out/r06prof_c5.log 2>&1
# the driver's own command: headline + alt precision + whole generation + configs[3] / configs[4] legs + cpu baseline
( time python bench.py --steps 20 --warmup 5 > gpurun_out/r06lines/bench_default.json 2> gpurun_out/r06lines/bench_default.err ) 2> gpurun_out/r06lines/bench_default.time
python bench.py --gpus 2 --single-device-test --steps 20 --warmup 3 > gpurun_out/r06lines/bench_2rank_selflaunched.json 2> gpurun_out/r06lines/bench_2rank.err
python bench.py --gpus 8 --single-device-test --games 128 --steps 10 --warmup 2 > gpurun_out/r06lines/bench_8rank_selflaunched.json 2> gpurun_out/r06lines/bench_8rank.err
for f in gpurun_out/r06lines/*.json; do echo "$f: $(tail -1 $f | cut -c1-160)"; done
cat gpurun_out/r06lines/bench_default.time
python -c "import __graft_entry__ as g; g.smoke(); print('smoke OK')" > gpurun_out/r06lines/smoke.log 2>&1; tail -1 gpurun_out/r06lines/smoke.log
