#!/bin/bash
# dev: build a variant of libpnerf_hip.so into tools/_build/<name>.so with extra defines (never the shipped build)
#   tools/build_variant.sh trace -DPN_PHASE_TRACE
set -e
cd "$(dirname "$0")/.."
NAME=$1; shift
D=tools/_build/var_$NAME; mkdir -p $D
cp pointnerf_amd/csrc/*.hip pointnerf_amd/csrc/*.h pointnerf_amd/csrc/Makefile $D/
sed -i 's#../../include#../../../include#g; s#^OUT   := ../libpnerf_hip.so#OUT   := ../'$NAME'.so#' $D/Makefile
make -C $D -j8 EXTRA_DEFS="$*" 2>&1 | grep -E "error|warning: v|Error" || true
ls -la tools/_build/$NAME.so
