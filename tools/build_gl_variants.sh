#!/bin/bash
# Builds variants of libsstts_hip.so that differ only in -D switches of griffin_lim.hip (tools-only ablations):
#   bash tools/build_gl_variants.sh NAME1:"-DFLAG -DFLAG2" NAME2:"..."   ->  tools/bin/lib_NAME.so
# Select one at run time with SSTTS_HIP_LIB=tools/bin/lib_NAME.so.
set -e
R=$(cd $(dirname $0)/.. && pwd)
B=$R/single-speaker-tts_amd/build
mkdir -p $R/tools/bin
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}
  ( hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -w -fno-slp-vectorize $flags -c $R/single-speaker-tts_amd/csrc/griffin_lim.hip -o $R/tools/bin/gl_$name.o \
    && hipcc --offload-arch=gfx950 -shared -fPIC -o $R/tools/bin/lib_$name.so $B/gemm_f32.o $B/cbhg_tail.o $B/gru.o $B/decoder.o $B/decoder_persistent.o $B/decoder_ws.o $R/tools/bin/gl_$name.o $B/griffin_lim_generic.o $B/reserve.o $B/api_handle.o $B/api_stages.o $B/api_pipeline.o \
    && echo built $name ) &
done
wait
