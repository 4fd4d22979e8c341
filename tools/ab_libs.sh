#!/bin/bash
# A/B builds of the HIP library in one GPU session: tools/ab_libs.sh lib1.so lib2.so ...
# ("default" = the in-tree build).  Prints the shapes_bench table per library.
for lib in "$@"; do
    echo "##### $lib"
    if [ "$lib" = default ]; then
        timeout -k 10 120 python tools/shapes_bench.py 2 4 2>&1 | grep -v amdgpu.ids || exit 1
    else
        SHARDMERGE_HIP_LIB=$PWD/$lib timeout -k 10 120 python tools/shapes_bench.py 2 4 2>&1 | grep -v amdgpu.ids || exit 1
    fi
done
