#!/bin/bash
mkdir -p gpurun_out
python scripts/quick_sw_tile64.py > gpurun_out/r06_sw_tile64.log 2>&1; cat gpurun_out/r06_sw_tile64.log
