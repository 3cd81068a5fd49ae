#!/bin/bash
# tools/build_variant.sh NAME "-DFLAG ..." : profiling build of libglf with extra flags -> tools/dbg/libglf_NAME.so
set -e
cd "$(dirname "$0")/.."
name=$1; flags=$2
P=image-processing-graph-laplacian_amd; C=$P/csrc
F="-O3 -std=c++17 --offload-arch=gfx950 -fPIC -ffp-contract=fast -fno-slp-vectorize -Iinclude -I$C $flags"
mkdir -p tools/dbg/obj_$name
for f in ctx affinity eigen nystroem filter pipeline comm nlm balance; do
  if [ $f = nystroem ] || [ ! -f tools/dbg/obj_$name/$f.o ]; then /opt/rocm/bin/hipcc $F -c $C/$f.hip -o tools/dbg/obj_$name/$f.o & fi
done; wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o tools/dbg/libglf_$name.so tools/dbg/obj_$name/*.o $C/host_util.o $P/host/png_codec.o -lz -pthread -ldl
