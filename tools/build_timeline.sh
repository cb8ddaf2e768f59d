#!/bin/bash
# tools/bin/lib_tl.so: the library with -DGL_TIMELINE in griffin_lim.hip and api_stages.hip (per-wave stamps of workgroup 0 and the
# start / end spread of all workgroups of the last Griffin-Lim iteration, printed by the next call); select it with SSTTS_HIP_LIB
set -e
R=$(cd $(dirname $0)/.. && pwd); B=$R/single-speaker-tts_amd/build; mkdir -p $R/tools/bin
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -w -fno-slp-vectorize -DGL_FAST_BUILD -DGL_TIMELINE $1 -c $R/single-speaker-tts_amd/csrc/griffin_lim.hip -o $R/tools/bin/gl_tl.o
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -w -DGL_TIMELINE -c $R/single-speaker-tts_amd/csrc/api_stages.hip -o $R/tools/bin/api_tl.o
hipcc --offload-arch=gfx950 -shared -fPIC -Wl,--no-undefined -o $R/tools/bin/lib_tl.so $B/gemm_f32.o $B/cbhg_tail.o $B/gru.o $B/decoder.o $B/decoder_persistent.o $B/decoder_ws.o $R/tools/bin/gl_tl.o $B/griffin_lim_generic.o $B/reserve.o $B/api_handle.o $R/tools/bin/api_tl.o $B/api_pipeline.o
echo built tools/bin/lib_tl.so
