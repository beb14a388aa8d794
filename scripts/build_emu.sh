#!/bin/bash
# the kernel emulation and the host code over it (what the CPU suite's `emu` fixture builds on demand): tests/emu/libpm_emu.so,
# parsnp_core_emu, libparsnp_core_emu.so -- built to .new files and moved into place, so that runs in flight keep their binary
set -e
cd "$(dirname "$0")/.."
E=tests/emu; H=parsnp_amd/csrc/host
g++ -O2 -std=c++17 -shared -fPIC -w -DPM_WAVE_EVENTS=5 $E/engine_emu.cpp -o $E/libpm_emu.so.new && mv $E/libpm_emu.so.new $E/libpm_emu.so
SRC=$(ls $H/*.cpp | grep -v -e capi.cpp -e merge_main.cpp)
g++ -O3 -mavx2 -std=c++17 -fopenmp -w -DPARSNP_TEST_HOOKS $SRC -L$E -lpm_emu -Wl,-rpath,'$ORIGIN' -o $E/parsnp_core_emu.new && mv $E/parsnp_core_emu.new $E/parsnp_core_emu
SRC=$(ls $H/*.cpp | grep -v -e main.cpp -e merge_main.cpp)
g++ -O3 -mavx2 -std=c++17 -fopenmp -fPIC -shared -w -DPARSNP_TEST_HOOKS $SRC -L$E -lpm_emu -Wl,-rpath,'$ORIGIN' -o $E/libparsnp_core_emu.so.new && mv $E/libparsnp_core_emu.so.new $E/libparsnp_core_emu.so
echo emu built
