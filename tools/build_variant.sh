#!/bin/bash
# Build an experimental variant of the HIP library next to the in-tree one:
#   tools/build_variant.sh NAME "-DFLAG1 -DFLAG2=3" [plan indices ... | main | side_N]
# Only the listed translation units are recompiled with the flags (default: 3 6 11 = the 8192-,
# 28672- and 7168-point plans of the Llama-3-70B shapes); every other object comes from the
# in-tree build (run `make -C shardmerge_amd/csrc -j8` first).  Result: exp_libs/lib_NAME.so
# (git-ignored, travels with gpurun); use with SHARDMERGE_HIP_LIB=... (tools/ab_kprof.sh).
set -e
NAME=$1; FLAGS=$2; shift 2
UNITS=${@:-3 6 11}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
SRC=$ROOT/shardmerge_amd/csrc
OUT=$ROOT/exp_libs; mkdir -p $OUT/obj_$NAME
BASE="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-variable -Wno-unused-but-set-variable"
noslp="0 1 4 5 6 7 8 9 10 11 12 13 14 15 18 19"
objs=""
pids=""
for u in $UNITS; do
    case $u in
        main) cmd="/opt/rocm/bin/hipcc $BASE $FLAGS -c $SRC/smhip_hip.hip -o $OUT/obj_$NAME/main.o" ;;
        side_*) g=${u#side_}; cmd="/opt/rocm/bin/hipcc $BASE $FLAGS -DSM_SIDE_GROUP=$g -c $SRC/smhip_side.hip -o $OUT/obj_$NAME/side_$g.o" ;;
        *) pf=""; for n in $noslp; do [ "$n" = "$u" ] && pf="-fno-slp-vectorize"; done
           cmd="/opt/rocm/bin/hipcc $BASE $pf $FLAGS -DSM_PLAN_INDEX=$u -c $SRC/smhip_inst.hip -o $OUT/obj_$NAME/inst_$u.o" ;;
    esac
    $cmd & pids="$pids $!"
done
for p in $pids; do wait $p; done
for f in $SRC/build/main.o $SRC/build/side_[0-9]*.o $SRC/build/inst_*.o; do
    b=$(basename $f)
    if [ -f $OUT/obj_$NAME/$b ]; then objs="$objs $OUT/obj_$NAME/$b"; else objs="$objs $f"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs -o $OUT/lib_$NAME.so
echo "built $OUT/lib_$NAME.so ($UNITS with: $FLAGS)"
