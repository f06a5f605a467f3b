#!/bin/bash
# tools/build_variant.sh <name> [-DFLAG=..]...   ->  multi_modal_normative_modeling_amd/libnmhip_<name>.so  (A/B experiments)
NAME=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -Iinclude "$@" \
  multi_modal_normative_modeling_amd/csrc/nmhip.hip multi_modal_normative_modeling_amd/csrc/nm_metrics.hip multi_modal_normative_modeling_amd/csrc/nm_prep.hip \
  -o multi_modal_normative_modeling_amd/libnmhip_$NAME.so 2>&1 | grep -E "error" -A3
ls -la multi_modal_normative_modeling_amd/libnmhip_$NAME.so
