#!/bin/bash
# dev only: build whole-library variants that differ in the flags of pp_ffn_dma.hip into scripts/micro/build/lib_<tag>.so:
#   ffd_variants.sh tag1 "-DFFD_X=1" tag2 "-DFFD_Y=2" ...        (then scripts/micro/ffd_variants_run.sh "tag1 tag2" on the GPU box)
set -e
here="$(dirname "$(readlink -f "$0")")"
mkdir -p "$here/build"
cd "$here/../../probpose_code_amd/csrc"
make -s -j8 >/dev/null
objs=$(ls build/*.o | grep -v pp_ffn_dma.o)
while [ $# -ge 2 ]; do
  tag=$1; flags=$2; shift 2
  ( /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-slp-vectorize -I../../include $flags -c pp_ffn_dma.hip -o "$here/build/ffd_$tag.o" \
      -Rpass-analysis=kernel-resource-usage 2>&1 | grep -A9 "proj_ffn_dma_kernel" | grep -E "VGPRs:|VGPRs Spill|ScratchSize" | tr '\n' ' '; echo " <- $tag ($flags)"
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$here/build/lib_$tag.so" $objs "$here/build/ffd_$tag.o" ) &
done
wait
