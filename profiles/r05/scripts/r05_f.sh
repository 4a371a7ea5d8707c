#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r05_f
mkdir -p $OUT
cd $R
timeout -k 10 400 python tools/ab_two_libs.py new=volumetricterrain_amd/libvtmc.so c1=tools/_ab/libvtmc_r05c1.so prev=tools/_ab/libvtmc_prev.so -- base emit_once=0 indexed=1 --rounds 11 > $OUT/ab_three_libs.txt 2>&1
cat $OUT/ab_three_libs.txt
timeout -k 10 400 python tools/ab_two_libs.py new=volumetricterrain_amd/libvtmc.so prev=tools/_ab/libvtmc_prev.so -- base emit_once=0 --rounds 11 > $OUT/ab_two_libs.txt 2>&1
cat $OUT/ab_two_libs.txt
