#!/bin/bash
# A library that differs from the in-tree build in ONE translation unit (other source, or other -D switches):
#   bash tools/build_variant.sh NAME decoder_persistent.hip "-DPD_KB=2" [path/to/other/source.hip]  ->  tools/bin/lib_NAME.so
# Select it at run time with SSTTS_HIP_LIB=tools/bin/lib_NAME.so.
set -e
R=$(cd $(dirname $0)/.. && pwd)
B=$R/single-speaker-tts_amd/build
name=$1; unit=$2; flags=$3; src=${4:-$R/single-speaker-tts_amd/csrc/$unit}
mkdir -p $R/tools/bin
extra=""
[ "$unit" = "griffin_lim.hip" -o "$unit" = "griffin_lim_generic.hip" ] && extra="-fno-slp-vectorize"
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -w $extra $flags -I$R/single-speaker-tts_amd/csrc -c $src -o $R/tools/bin/${name}_unit.o
objs=""
for u in gemm_f32 cbhg_tail gru decoder decoder_persistent decoder_ws griffin_lim griffin_lim_generic reserve api_handle api_stages api_pipeline; do
  if [ "$u.hip" = "$unit" ]; then objs="$objs $R/tools/bin/${name}_unit.o"; else objs="$objs $B/$u.o"; fi
done
hipcc --offload-arch=gfx950 -shared -fPIC -Wl,--no-undefined -o $R/tools/bin/lib_$name.so $objs
echo built tools/bin/lib_$name.so
