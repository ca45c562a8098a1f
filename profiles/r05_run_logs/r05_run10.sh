#!/bin/bash
O=gpurun_out/r05_23; mkdir -p $O
timeout 300 python scripts/diag_persist_stamps.py 1 > $O/stamps_r1.txt 2>&1; tail -24 $O/stamps_r1.txt
