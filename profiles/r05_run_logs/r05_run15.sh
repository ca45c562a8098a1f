#!/bin/bash
O=gpurun_out/r05_29; mkdir -p $O
timeout 300 python scripts/diag_driver_regime.py > $O/out.txt 2>&1; tail -60 $O/out.txt
