#!/bin/bash
# Tuning experiments: the same library under other -D switches, side by side with the product build.
#   tools/build_variant.sh <name> [-DFOO=1 ...]   ->  freddie_amd/libfreddie_seg_<name>.so   (use: FSEG_LIB=$PWD/freddie_amd/libfreddie_seg_<name>.so)
N=$1; shift
cd "$(dirname "$0")/.." && python -m freddie_amd.build --variant "$N" "$@"
