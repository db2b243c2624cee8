#!/bin/bash
# Runs the VALU rate micro-benchmark (tools/valu_peak, built beforehand) with 8, 2 and 1 wavefronts per SIMD.
# usage: tools/valu_peak_run.sh <outdir>
OUT=$1; mkdir -p $OUT
timeout -k 10 120 $GRAFT_REPO_ROOT/tools/valu_peak 2048 > $OUT/valu_peak_8waves.txt 2>&1 &&
timeout -k 10 120 $GRAFT_REPO_ROOT/tools/valu_peak 512 > $OUT/valu_peak_2waves.txt 2>&1 &&
timeout -k 10 120 $GRAFT_REPO_ROOT/tools/valu_peak 256 > $OUT/valu_peak_1wave.txt 2>&1
