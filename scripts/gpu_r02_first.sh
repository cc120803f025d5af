#!/bin/bash
# first GPU visit of round 2: tests, smoke, bench (default + driver's command), probe
export TMPDIR=/tmp
OUT=gpurun_out/r02a
rm -rf $OUT; mkdir -p $OUT
(time timeout 900 python -m pytest tests -m gpu -x -q --durations=8) > $OUT/pytest.log 2>&1; tail -5 $OUT/pytest.log
timeout 300 python __graft_entry__.py smoke > $OUT/smoke.log 2>&1; tail -2 $OUT/smoke.log
timeout 600 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; tail -c 1500 $OUT/bench_default.json
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_k20.json 2> $OUT/bench_k20.err; tail -c 600 $OUT/bench_k20.json
timeout 900 python scripts/r02_probe.py all > $OUT/probe.log 2>&1; cat $OUT/probe.log
