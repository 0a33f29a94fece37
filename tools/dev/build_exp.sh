#!/bin/bash
# Experiment build of the library: tools/dev/build_exp.sh NAME "<extra hipcc flags>" [units...]  ->  gigalens_amd/lib/exp/NAME.so
# Recompiles the given translation units (default: the LL_GRAD generic unit) with the extra flags and links them with the
# shipped build's other objects (build/*.o must be current: run __graft_entry__.build() first).  Never tracked, never shipped.
set -e
NAME=$1; FLAGS=$2; shift 2
UNITS=${@:-gl_generic_noslp_mode3}
R=$(cd "$(dirname "$0")/../.." && pwd)
mkdir -p $R/gigalens_amd/lib/exp $R/build/exp
OBJS=""
for o in $R/build/*.o; do
  u=$(basename $o .o); skip=0
  for x in $UNITS; do [ "$x" = "$u" ] && skip=1; done
  [ $skip = 0 ] && OBJS="$OBJS $o"
done
for u in $UNITS; do
  extra=""; case $u in *noslp*) extra="-fno-slp-vectorize";; esac
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I $R/include -I $R/build $extra $FLAGS -c $R/gigalens_amd/csrc/$u.hip -o $R/build/exp/${NAME}_$u.o &
  OBJS="$OBJS $R/build/exp/${NAME}_$u.o"
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/gigalens_amd/lib/exp/$NAME.so $OBJS -lhiprtc
echo built gigalens_amd/lib/exp/$NAME.so
