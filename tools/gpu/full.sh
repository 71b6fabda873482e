#!/bin/bash
# the whole GPU suite + smoke, as the driver runs them
cd "$(dirname "$0")/../.."; o=gpurun_out
timeout 2400 python -m pytest tests/ -x -q -m gpu > $o/full_tests.log 2>&1; echo "pytest rc=$?" >> $o/full_tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" >> $o/full_tests.log 2>&1
