#!/bin/bash
# Tuning aid: builds ab_libs/libresynth_<name>.so for each "name:extra hipcc flags" argument (ra_synth.hip and ra_analysis.hip are recompiled with the flags).
# Select one at run time with RESYNTH_AMD_LIB=ab_libs/libresynth_<name>.so.
set -e
cd "$(dirname "$0")/../cpp_audio_amd/csrc"
make -s
mkdir -p ../../ab_libs
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -w"
for spec in "$@"; do
  name="${spec%%:*}"; extra="${spec#*:}"
  /opt/rocm/bin/hipcc $FLAGS $extra -c ra_synth.hip -o /tmp/ra_synth_$name.o &
  /opt/rocm/bin/hipcc $FLAGS $extra -c ra_analysis.hip -o /tmp/ra_analysis_$name.o
  wait
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../ab_libs/libresynth_$name.so ra_capi.o /tmp/ra_analysis_$name.o ra_track.o /tmp/ra_synth_$name.o ra_vocoder.o ra_carrier.o ra_post.o ra_voice.o ra_notes.o
  echo "built $name ($extra)"
done
