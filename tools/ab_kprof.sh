#!/bin/bash
# per-kernel times of a few shapes for several builds of the library: tools/ab_kprof.sh default exp_libs/lib_x.so ...
# SHAPES="rows cols k;rows cols k" overrides the shape list
R=$GRAFT_REPO_ROOT
cd $R
SHAPES=${SHAPES:-"28672 8192 3;8192 8192 2;8192 28672 3"}
for lib in "$@"; do
  echo "##### $lib"
  IFS=';' read -ra SH <<< "$SHAPES"
  for s in "${SH[@]}"; do
    if [ "$lib" = default ]; then python3 tools/kprof.py $s 2>&1; else SHARDMERGE_HIP_LIB=$R/$lib python3 tools/kprof.py $s 2>&1; fi | grep -E "^\[|f1_rows|f2s_cols|f2_cols|i1_cols|i2_rows|aten_norm_part|blend|select_lvl2 |spec_rescale"
  done
done
