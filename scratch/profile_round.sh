#!/bin/bash
# Measurements behind profiles/r01/v6_*: bench JSON, rocprofv3 kernel stats, HBM traffic (two PMC passes), SQ counters.
# Counters are collected in their own runs (no trace domains beside them).
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/v6; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
python3 $R/bench.py > $O/bench.json 2> $O/bench.err; echo "bench done"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --no-cpu-baseline --no-profile --no-extras > $O/stats.log 2>&1; echo "stats done"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-profile --no-extras > $O/pmc_fetch.log 2>&1; echo "fetch done"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-profile --no-extras > $O/pmc_write.log 2>&1; echo "write done"
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY" \
           "SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INST_LEVEL_VMEM"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $O/sq/p$i -- python3 $R/scratch/fullwidth_run.py 12 > $O/sq_p$i.log 2>&1; echo "sq pass $i done"
done
