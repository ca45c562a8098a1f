#!/bin/bash
O=gpurun_out/r05_32; mkdir -p $O
timeout 600 python scripts/ab_stats_resident.py > $O/ab.txt 2>&1; tail -30 $O/ab.txt
