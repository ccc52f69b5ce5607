#!/bin/bash
# usage (GPU box): tools/c4_variants.sh  -> the 4096-instance 4K realtime frame (tools/profile_c4.py) under the two-level experiments of round 4
for rep in 1 2; do
for v in "default 0" "si 0" "default 1" "si 1"; do
  set -- $v
  if [ "$1" = default ]; then unset DXR_AMD_LIB; else export DXR_AMD_LIB=$PWD/dxrexperiments_amd/lib/variants/lib$1.so; fi
  echo "lib=$1 RT_PRIMARY_PERSISTENT=$2: $(RT_DEBUG_OPTIONS=primary_persistent=$2 python tools/profile_c4.py 8 2>/dev/null | head -2 | tr '\n' ' ')"
done; done
