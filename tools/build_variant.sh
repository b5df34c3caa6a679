#!/bin/bash
# A measurement build of the library under another name: tools/build_variant.sh NAME "-DFOO=1 -DBAR=2"
# -> deeprl_signal_control_amd/libtsc_NAME.so (git-ignored; select it with TSC_LIB=<path> for an A/B run, never for a reported figure)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; DEFS=${2:-}
OBJ=$(mktemp -d)
for f in $ROOT/deeprl_signal_control_amd/csrc/*.hip; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wno-unused-function $DEFS -I $ROOT/include -c $f -o $OBJ/$(basename $f .hip).o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/deeprl_signal_control_amd/libtsc_$NAME.so $OBJ/*.o
rm -rf $OBJ
echo $ROOT/deeprl_signal_control_amd/libtsc_$NAME.so
