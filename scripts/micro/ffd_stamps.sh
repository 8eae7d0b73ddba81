#!/bin/bash
# dev only: build the library with -DFFD_STAMP=1 (here, no GPU needed) and, on the GPU box, print the phase breakdown of the paired proj + FFN kernel:
#   scripts/micro/ffd_stamps.sh build [2]   |   gpurun -- 'bash scripts/micro/ffd_stamps.sh run [fine]'      (2 / fine: the LayerNorm phases in pieces)
here="$(dirname "$(readlink -f "$0")")"; root="$here/../.."
if [ "$1" = build ]; then
  cd "$root/probpose_code_amd/csrc" && make -s -j8 >/dev/null && mkdir -p "$here/build" &&
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-slp-vectorize -I../../include -DFFD_STAMP=${2:-1} -c pp_ffn_dma.hip -o "$here/build/ffd_stamp.o" 2>/dev/null &&
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$here/build/lib_stamp.so" $(ls build/*.o | grep -v pp_ffn_dma.o) "$here/build/ffd_stamp.o" && echo built
else
  cp "$root/probpose_code_amd/libprobpose_mi355x.so" /tmp/lib_orig.so
  cp "$here/build/lib_stamp.so" "$root/probpose_code_amd/libprobpose_mi355x.so"
  python "$here/ffd_stamps.py" $2
  cp /tmp/lib_orig.so "$root/probpose_code_amd/libprobpose_mi355x.so"
fi
