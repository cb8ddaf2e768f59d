#!/bin/bash
# Experiment build: libsstts_hip.so whose griffin_lim device code went through tools/strip_nops.py.
#   bash tools/build_gl_stripped.sh NAME "-DFLAGS"   ->  tools/bin/lib_NAME.so
set -e
R=$(cd $(dirname $0)/.. && pwd)
B=$R/single-speaker-tts_amd/build
L=/opt/rocm/lib/llvm/bin
name=$1; flags=$2
W=$R/tools/bin/strip_$name; mkdir -p $W
CF="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -w -fno-slp-vectorize $flags"
hipcc $CF -S --cuda-device-only $R/single-speaker-tts_amd/csrc/griffin_lim.hip -o $W/dev.s
python3 $R/tools/strip_nops.py $W/dev.s $W/dev_stripped.s
$L/clang -x assembler -target amdgcn-amd-amdhsa -mcpu=gfx950 -c $W/dev_stripped.s -o $W/dev.o
$L/lld -flavor gnu -m elf64_amdgpu --no-undefined -shared $W/dev.o -o $W/dev.out
$L/clang-offload-bundler -type=o -bundle-align=4096 -targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--gfx950 -input=/dev/null -input=$W/dev.out -output=$W/dev.hipfb
hipcc $CF --cuda-host-only -Xclang -fcuda-include-gpubinary -Xclang $W/dev.hipfb -c $R/single-speaker-tts_amd/csrc/griffin_lim.hip -o $W/gl.o
hipcc --offload-arch=gfx950 -shared -fPIC -o $R/tools/bin/lib_$name.so $B/gemm_f32.o $B/cbhg_tail.o $B/gru.o $B/decoder.o $B/decoder_persistent.o $B/decoder_ws.o $W/gl.o $B/griffin_lim_generic.o $B/reserve.o $B/api_handle.o $B/api_stages.o $B/api_pipeline.o
echo built $name
