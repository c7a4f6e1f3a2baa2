#!/bin/bash
# Experiment build of the library next to the shipping one: tools/build_variant.sh x1 "-DPARO_E2_CONS=7" [object ...]
#   copies paroquant_amd/_lib to paroquant_amd/_lib_<name>, drops the named objects (default: engine2.o) and rebuilds them with the extra
#   flags; A/B on the GPU box with PARO_LIB_DIR=_lib_<name>.  (make does not track flags: only the dropped objects see them.)
set -e
NAME=$1; EXTRA=$2; shift 2 || true
OBJS=${@:-engine2.o}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/paroquant_amd/_lib_$NAME
rm -rf $OUT; mkdir -p $OUT
cp -p $ROOT/paroquant_amd/_lib/*.o $OUT/
for o in $OBJS; do rm -f $OUT/$o; done
make -s -C $ROOT/paroquant_amd/csrc OUT=$OUT EXTRA="$EXTRA" 2>&1 | grep -v "^/opt/rocm/bin/hipcc" || true
rm -f $OUT/*.o
ls -la $OUT/libparo_mi355x.so
