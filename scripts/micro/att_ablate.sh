#!/bin/bash
# dev only: pp_attention.hip with compile-time ablations (-DATT_ABL=1 no V transpose, 2 no math, 3 no K / V loads) into
# scripts/micro/build/libatt_abl<N>.so; time them with `python scripts/micro/att_abl_bench.py` on the GPU box.
set -e
cd "$(dirname "$(readlink -f "$0")")/../../probpose_code_amd/csrc"
mkdir -p ../../scripts/micro/build
for d in 1 2 3; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../../include -DATT_ABL=$d -shared pp_attention.hip pp_api.hip -o ../../scripts/micro/build/libatt_abl$d.so
done
