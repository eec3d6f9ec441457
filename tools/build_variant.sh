#!/bin/bash
# A/B builds of the kernel library: bash tools/build_variant.sh NAME "extra hipcc flags" [files...]
# -> acai_omr_amd/csrc/variants/NAME.so (git-ignored, ships to the GPU box); select it with ACAI_OMR_LIB=<path>.
# The named files (default: the attention sources) are compiled with the extra flags, every other object comes from csrc/build/.
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
C=$R/acai_omr_amd/csrc
NAME=$1; EXTRA=$2; shift 2 || true
FILES=${@:-attn_varlen.hip attn_bwd.hip}
python3 -c "import sys; sys.path.insert(0, '$R'); from acai_omr_amd import _lib; _lib.build()" >/dev/null
V=$C/variants/$NAME; mkdir -p $V
OBJS=""
for f in $C/*.hip; do
  b=$(basename $f)
  if echo " $FILES " | grep -q " $b "; then
    hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -Wno-unused-value $EXTRA -c $f -o $V/$b.o &
    OBJS="$OBJS $V/$b.o"
  else
    OBJS="$OBJS $C/build/$b.o"
  fi
done
wait
hipcc --offload-arch=gfx950 -shared -fPIC -o $C/variants/$NAME.so $OBJS
echo $C/variants/$NAME.so
