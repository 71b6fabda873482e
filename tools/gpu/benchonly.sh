#!/bin/bash
cd "$(dirname "$0")/../.."; o=gpurun_out
timeout 900 python bench.py > $o/bench_only.log 2> $o/bench_only.err; tail -1 $o/bench_only.log > $o/bench_only.json
