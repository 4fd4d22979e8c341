#!/bin/bash
# A/B builds of the HIP library on the default bench (8 streams): tools/ab_bench.sh lib1.so lib2.so ... (two rounds)
for round in 1 2; do
for lib in "$@"; do
    if [ "$lib" = default ]; then L=""; else L="$PWD/$lib"; fi
    v=$(SHARDMERGE_HIP_LIB=$L timeout -k 10 200 python bench.py --no-cpu-baseline --no-profile 2>/dev/null | grep -o '"value": [0-9.]*')
    echo "$lib $v"
done
done
