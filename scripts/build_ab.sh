#!/bin/bash
# Alternative build of librtts_hip.so for kernel A/B runs on one box: scripts/build_ab.sh <name> [extra hipcc flags...]
# -> reformer-tts_amd/lib/librtts_<name>.so (select it with RTTS_LIB=...)
set -e
cd "$(dirname "$0")/.."
name=$1; shift
out=reformer-tts_amd/lib/ab_$name
mkdir -p $out
for f in reformer-tts_amd/csrc/*.hip reformer-tts_amd/csrc/*.cpp; do
  extra=""
  [ "$(basename $f)" = "lsh_attn_bwd.hip" ] && extra="-mllvm -amdgpu-sched-strategy=iterative-ilp"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-value -ffp-contract=on -fno-slp-vectorize $extra "$@" -x hip -c $f -o $out/$(basename $f).o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o reformer-tts_amd/lib/librtts_$name.so $out/*.o
echo reformer-tts_amd/lib/librtts_$name.so
