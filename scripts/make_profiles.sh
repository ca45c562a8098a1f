#!/bin/bash
# Regenerates the round's profile artifacts on the GPU box (run through gpurun from the repo root):
#   bash scripts/make_profiles.sh r06
# writes gpurun_out/<tag>_*; copy what is to be judged into profiles/.
TAG=${1:-r06}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out
mkdir -p $OUT /tmp/w
cd /tmp && export TMPDIR=/tmp
# the bench lines: the defaults, and the driver's arguments
python3 $R/bench.py > $OUT/${TAG}_bench_default.json 2> $OUT/${TAG}_bench_default.err
python3 $R/bench.py --steps 20 --warmup 5 > $OUT/${TAG}_bench_driver_args.json 2> $OUT/${TAG}_bench_driver_args.err
# kernel trace of the headline step (reduction 10, then 1): no events on the stream, fresh rows
for r in 10 1; do
  rm -rf /tmp/w/kt$r
  timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/w/kt$r -o t -- python3 $R/bench.py --reduction $r --steps 600 --warmup 400 --no-cpu-baseline --steady-steps 0 --no-breakdown > /tmp/w/kt$r.log 2>&1
  DB=$(find /tmp/w/kt$r -name "*.db" | head -1)
  { echo "rocprofv3 --kernel-trace --stats -- python3 bench.py --reduction $r --steps 600 --warmup 400 --no-cpu-baseline --steady-steps 0 --no-breakdown   (second half of the trace)"; python3 $R/scripts/prof_summary.py $DB 0.5; } > $OUT/${TAG}_kernel_trace_bench_r$r.txt 2>&1
  python3 $R/scripts/step_timeline.py $DB 1 > $OUT/${TAG}_step_timeline_r$r.txt 2>&1
done
# HBM traffic: FETCH_SIZE and WRITE_SIZE in separate passes
DBS=""
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/w/pmc_$c; timeout 600 rocprofv3 --kernel-trace --pmc $c -d /tmp/w/pmc_$c -o t -- python3 $R/bench.py --steps 100 --warmup 40 --no-cpu-baseline --steady-steps 0 --no-breakdown > /tmp/w/pmc_$c.log 2>&1
  DBS="$DBS $(find /tmp/w/pmc_$c -name '*.db' | head -1)"
done
python3 $R/scripts/pmc_summary.py $DBS --json $OUT/${TAG}_pmc_hbm_traffic.json > $OUT/${TAG}_pmc_hbm_traffic.txt 2>&1
SQC="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS"
rm -rf /tmp/w/sq; timeout 600 rocprofv3 --kernel-trace --pmc $SQC -d /tmp/w/sq -o t -- python3 $R/bench.py --steps 100 --warmup 40 --no-cpu-baseline --steady-steps 0 --no-breakdown > /tmp/w/sq.log 2>&1
{ echo "rocprofv3 --kernel-trace --pmc $SQC"; echo "(averages over the counter-instance rows of all dispatches; SQ_* cycle counters are in quad-cycles; use the RATIOS)"; echo; python3 $R/scripts/pmc_summary.py $(find /tmp/w/sq -name '*.db' | head -1); } > $OUT/${TAG}_pmc_sq_counters.txt 2>&1
# config 5's per-GPU shape (p = 200 000, k = 256, b = 256, reduction 12): section times + kernel trace
python3 $R/scripts/diag_hcp_shape.py > $OUT/${TAG}_c5_shape_sections.txt 2>&1
rm -rf /tmp/w/c5; timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/w/c5 -o t -- python3 $R/scripts/diag_hcp_shape.py > /tmp/w/c5.log 2>&1
{ tail -8 /tmp/w/c5.log; python3 $R/scripts/prof_summary.py $(find /tmp/w/c5 -name "*.db" | head -1) 0.3; } > $OUT/${TAG}_c5_shape_kernel_trace.txt 2>&1
ls -la $OUT | grep $TAG
# the multi-GPU (two-phase) step's cost on ONE GPU: the fused step against the forced reduction (world = 1: the all-reduce
# is the identity, its launch and stream ordering are real) on the library's own RCCL communicator (one call per chunk,
# the default of `bench.py --gpus N`) and on torch.distributed (two calls + a collective per minibatch)
for v in fused native torch; do
  case $v in
    fused) F="" ;;
    native) F="--force-reduce --native-rccl" ;;
    torch) F="--force-reduce --torch-collective" ;;
  esac
  export MASTER_PORT=$((20000 + RANDOM % 20000))
  python3 $R/bench.py --steps 600 --warmup 400 --no-cpu-baseline --steady-steps 0 --no-breakdown $F > $OUT/${TAG}_two_phase_$v.json 2> /tmp/w/tp_$v.err
  rm -rf /tmp/w/tp_$v
  export MASTER_PORT=$((20000 + RANDOM % 20000))
  timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/w/tp_$v -o t -- python3 $R/bench.py --steps 600 --warmup 400 --no-cpu-baseline --steady-steps 0 --no-breakdown $F > /tmp/w/tp_$v.log 2>&1
  DB=$(find /tmp/w/tp_$v -name "*.db" | head -1)
  { echo "rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 600 --warmup 400 --no-cpu-baseline --steady-steps 0 --no-breakdown $F   (second half of the trace)"; python3 $R/scripts/prof_summary.py $DB 0.5; } > $OUT/${TAG}_two_phase_${v}_kernel_trace.txt 2>&1
  python3 $R/scripts/step_timeline.py $DB 1 > $OUT/${TAG}_two_phase_${v}_timeline.txt 2>&1
done
# BASELINE configs 2-4 and the reference's HCP run (C6): GPU throughput, the CPU oracle on a bounded prefix, the dominant
# section's roofline fraction
timeout 1500 python3 $R/tests/diag/bench_configs_full.py --only c2,c3,c4,c6 > $OUT/${TAG}_bench_configs.jsonl 2> $OUT/${TAG}_bench_configs.err
rm -rf /tmp/w/c6; timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/w/c6 -o t -- python3 $R/scripts/bench_configs.py --only c6 --c6-batches 5 > /tmp/w/c6.log 2>&1
{ tail -3 /tmp/w/c6.log; python3 $R/scripts/prof_summary.py $(find /tmp/w/c6 -name "*.db" | head -1) 0.3; } > $OUT/${TAG}_cfg_c6_kernel_trace.txt 2>&1
ls -la $OUT | grep $TAG
# BASELINE config 5 as a bench workload (p = 200 000, reduction 12): fused, and the two-phase step forced with one rank on
# the library's RCCL communicator / on torch's collective - what the head mirror + scatter of 17 MB costs before a byte crosses xGMI
for v in fused native torch; do
  case $v in
    fused) F="" ;;
    native) F="--force-reduce" ;;
    torch) F="--force-reduce --torch-collective" ;;
  esac
  export MASTER_PORT=$((20000 + RANDOM % 20000))
  python3 $R/bench.py --features 200000 --reduction 12 --steps 200 --warmup 60 --no-cpu-baseline $F > $OUT/${TAG}_two_phase_c5_$v.json 2> /tmp/w/tpc5_$v.err
done
export MASTER_PORT=$((20000 + RANDOM % 20000))
rm -rf /tmp/w/tpc5; timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/w/tpc5 -o t -- python3 $R/bench.py --features 200000 --reduction 12 --steps 100 --warmup 40 --no-cpu-baseline --no-breakdown --force-reduce > /tmp/w/tpc5.log 2>&1
{ echo "rocprofv3 --kernel-trace --stats -- python3 bench.py --features 200000 --reduction 12 --steps 100 --warmup 40 --no-cpu-baseline --no-breakdown --force-reduce (second half)"; python3 $R/scripts/prof_summary.py $(find /tmp/w/tpc5 -name "*.db" | head -1) 0.5; } > $OUT/${TAG}_two_phase_c5_native_kernel_trace.txt 2>&1
python3 $R/scripts/diag_persist_stamps.py 10 1 > $OUT/${TAG}_persist_stamps.txt 2>&1
# config 5 with the defaults of its workload (300 timed minibatches, the stream resident in HBM), the statistics product with
# the code matrix in registers on its own (scripts/micro/res_gemm.hip), and the SQ counters of the reduction-1 and
# config-5 steps (the three products of the round-4 review: MFMA-busy and LDS-wait shares per kernel)
python3 $R/bench.py --features 200000 --reduction 12 --no-cpu-baseline > $OUT/${TAG}_bench_c5.json 2> $OUT/${TAG}_bench_c5.err
{ for a in "10000" "200000" "4104 100 200"; do $R/scripts/micro/res_gemm $a; done; } > $OUT/${TAG}_stats_resident_micro.txt 2>&1
rm -rf /tmp/w/sq1; timeout 600 rocprofv3 --kernel-trace --pmc $SQC -d /tmp/w/sq1 -o t -- python3 $R/bench.py --reduction 1 --steps 100 --warmup 40 --no-cpu-baseline --steady-steps 0 --no-breakdown > /tmp/w/sq1.log 2>&1
{ echo "rocprofv3 --kernel-trace --pmc $SQC -- python3 bench.py --reduction 1 --steps 100 --warmup 40 ..."; echo "(SQ_* cycle counters are in quad-cycles; use the RATIOS)"; echo; python3 $R/scripts/pmc_summary.py $(find /tmp/w/sq1 -name '*.db' | head -1); } > $OUT/${TAG}_pmc_sq_counters_r1.txt 2>&1
rm -rf /tmp/w/sq5; timeout 600 rocprofv3 --kernel-trace --pmc $SQC -d /tmp/w/sq5 -o t -- python3 $R/bench.py --features 200000 --reduction 12 --steps 40 --warmup 20 --no-cpu-baseline --no-breakdown > /tmp/w/sq5.log 2>&1
{ echo "rocprofv3 --kernel-trace --pmc $SQC -- python3 bench.py --features 200000 --reduction 12 --steps 40 --warmup 20 ..."; echo "(SQ_* cycle counters are in quad-cycles; use the RATIOS)"; echo; python3 $R/scripts/pmc_summary.py $(find /tmp/w/sq5 -name '*.db' | head -1); } > $OUT/${TAG}_pmc_sq_counters_c5.txt 2>&1
ls -la $OUT | grep $TAG
# round 6: the masked minibatch (C4) - kernel trace, the in-kernel stamps of the one-launch variants, the A/B of the policies
rm -rf /tmp/w/c4; timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/w/c4 -o t -- python3 $R/scripts/bench_configs.py --only c4 --c4-batches 2000 > /tmp/w/c4.log 2>&1
{ tail -2 /tmp/w/c4.log; python3 $R/scripts/prof_summary.py $(find /tmp/w/c4 -name "*.db" | head -1) 0.0; } > $OUT/${TAG}_cfg_c4_kernel_trace.txt 2>&1
python3 $R/scripts/c4_timeline.py $(find /tmp/w/c4 -name "*.db" | head -1) > $OUT/${TAG}_cfg_c4_timeline.txt 2>&1
timeout 300 python3 $R/scripts/diag_c4_host.py > $OUT/${TAG}_cfg_c4_host_cprofile.txt 2>&1
# the measured values behind two tolerances of the suite
timeout 600 python3 -m pytest $R/tests/test_gpu_step.py -q -m gpu -s -k "gram_accumulator_out_of_range" 2>&1 | grep -a "accumulator vs records" > $OUT/${TAG}_accumulator_vs_records.txt
{ for v in 2 4; do timeout 300 python3 $R/scripts/diag_recsys_stamps.py $v; done; } > $OUT/${TAG}_recsys_fused_stamps.txt 2>&1
timeout 600 bash $R/scripts/ab_recsys_fused.sh 2>&1 | grep fused= > $OUT/${TAG}_ab_recsys_fused.txt
# the spread l1 projection at the HCP shape: stamps, and the A/B against the last workgroup's projection
# C6's per-atom launch: stamps of the last workgroup (register projection, the default) and of the spread projection (12=3);
# A/B of the projection routes (12 = 1 registers / 3 spread / 0 LDS scans) and of the riding gradient rows (14 = 1 / 0)
{ timeout 300 python3 $R/scripts/diag_atom_stamps_c6.py; timeout 300 python3 $R/scripts/diag_atom_stamps_c6.py 12=3; } > $OUT/${TAG}_atom_mwg_c6_stamps.txt 2>&1
{ for v in "12=1" "12=3" "12=0" "14=0" "12=1"; do echo "# --debug-set $v"; timeout 300 python3 $R/scripts/bench_configs.py --only c6 --debug-set $v 2>/dev/null; done; } > $OUT/${TAG}_ab_atom_mwg_c6.jsonl
# the sweep-flip census at 204 800 samples of both reductions (bench.py: flip_rate_block; ~15 min of CPU oracle; FLIP=0 skips it)
[ "${FLIP:-1}" = "1" ] && python3 - > $OUT/${TAG}_flip_census.json 2> $OUT/${TAG}_flip_census.err <<PY
import sys, json
sys.path.insert(0, "$R")
import torch, bench
out = [bench.flip_rate_block(r, 800, torch.device("cuda", 0), log=lambda m: sys.stderr.write(m + "\n")) for r in (10.0, 1.0)]
print(json.dumps(out))
PY
ls -la $OUT | grep $TAG
