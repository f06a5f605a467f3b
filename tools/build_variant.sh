#!/bin/bash
# tools/build_variant.sh <name> [-DFLAG=..]...   ->  multi_modal_normative_modeling_amd/libnmhip_<name>.so  (A/B experiments;
# load it with NMHIP_LIB_NAME=libnmhip_<name>.so).  The flags reach the three translation units that include nm_core.inc; the
# others are taken from build/ (run __graft_entry__.build() first).
NAME=$1; shift
C=multi_modal_normative_modeling_amd/csrc
O=build/var_$NAME
mkdir -p $O
for f in nmhip nm_rowsplit nm_devpass; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude "$@" -c $C/$f.hip -o $O/$f.o > $O/$f.log 2>&1 &
done
wait
grep -E "error" -A3 $O/*.log
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $O/nmhip.o $O/nm_rowsplit.o $O/nm_devpass.o build/nm_metrics.o build/nm_prep.o build/nm_fusion.o \
  -o multi_modal_normative_modeling_amd/libnmhip_$NAME.so
ls -la multi_modal_normative_modeling_amd/libnmhip_$NAME.so
