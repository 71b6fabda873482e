#!/bin/bash
OMG_PDIST_GATE_DEBUG=1 OMG_PDIST_GATE=1 PYTHONPATH=. timeout 120 python tools/pdist_loopback_time.py 8 2>&1 | grep -E "pdist gate|world|Error" | head -40 > gpurun_out/r05_gate_w8_dbg.txt
