#!/bin/bash
O=gpurun_out/r05_15; mkdir -p $O
MODL_DIAG_NO_RIDER=1 timeout 300 python scripts/diag_persist_stamps.py 10 > $O/stamps_norider.txt 2>&1
timeout 300 python scripts/diag_persist_stamps.py 10 > $O/stamps.txt 2>&1
head -5 $O/stamps_norider.txt; grep "row workgroup 0" $O/stamps_norider.txt; grep "block 3:" $O/stamps_norider.txt
