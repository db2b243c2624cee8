#!/bin/bash
# Runs the VALU rate micro-benchmark (tools/valu_peak, built beforehand) and samples the engine clock
# rocm-smi reports while it runs.  usage: tools/valu_peak_run.sh <outdir>
OUT=$1; mkdir -p $OUT
( for i in $(seq 1 12); do rocm-smi --showclocks 2>/dev/null | grep -i "sclk" >> $OUT/sclk_samples.txt; sleep 0.4; done ) &
SMI=$!
timeout -k 10 120 $GRAFT_REPO_ROOT/tools/valu_peak > $OUT/valu_peak.txt 2>&1
rc=$?
wait $SMI
exit $rc
