#!/bin/bash
python -m pytest tests -m gpu -x -q 2>&1 | tail -12 > gpurun_out/r05_full.log
