#!/bin/bash
# Collects SQ counters for a short bench run in separate rocprofv3 passes (run on the GPU box).
# usage: tools/pmc_collect.sh <outdir under gpurun_out> [bench args...]
# then:  python tools/pmc_summary.py profiles/r02_pmc_per_kernel.txt profiles/r02_pmc_per_kernel.json <bench batch, 512 by default> \
#            gpurun_out/<outdir>/p{1,2,3,4}_counter_collection.csv
set -e
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# counter passes: one frame group per batch, so that every launch of a kernel covers the whole batch
export ORBHIP_PARTS=1
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $OUT -o p1 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 1 --no-cpu --no-extras "$@" > $OUT/p1.log 2>&1 &&
timeout -s KILL 120 rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU --output-format csv -d $OUT -o p2 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 1 --no-cpu --no-extras "$@" > $OUT/p2.log 2>&1 &&
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT -o p3 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 1 --no-cpu --no-extras "$@" > $OUT/p3.log 2>&1 &&
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT -o p4 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 1 --no-cpu --no-extras "$@" > $OUT/p4.log 2>&1 || exit 1
# kernel trace, single group: the per-kernel averages are whole-batch launches, as in bench.py's stage timing
timeout -s KILL 120 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o kt1 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-cpu --no-extras "$@" > $OUT/kt1.log 2>&1 || exit 1
unset ORBHIP_PARTS
# kernel trace of the default run (two frame groups on two streams; launches of mixed size)
timeout -s KILL 120 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o kt -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-cpu --no-extras "$@" > $OUT/kt.log 2>&1
ls $OUT
