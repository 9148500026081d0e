#!/bin/bash
# The CPU suite on a sanitizer build of the emulator (host + device sources compiled by g++; device arrays are heap blocks, LDS is static storage there):
#   bash tools/emu_sanitize.sh address|undefined [pytest args]
# Builds into /tmp/emu_<kind>, swaps tests/emu/libdm_emu.so for the run and puts the normal library back afterwards.  The C-caller tests are
# deselected (a plain gcc link cannot resolve the sanitizer runtime).  Round 5: both kinds clean over the whole `-m "not gpu"` suite (NOTES.md).
set -e
KIND=${1:-address}; shift || true
ROOT=$(cd "$(dirname "$0")/.." && pwd); EMU=$ROOT/tests/emu
case $KIND in
  address)   FLAGS="-fsanitize=address"; RT=$(gcc -print-file-name=libasan.so); export ASAN_OPTIONS=detect_leaks=0:detect_stack_use_after_return=0:halt_on_error=1 ;;
  undefined) FLAGS="-fsanitize=undefined -fno-sanitize-recover=undefined"; RT=$(gcc -print-file-name=libubsan.so); export UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 ;;
  *) echo "usage: $0 address|undefined [pytest args]"; exit 2 ;;
esac
make -s -C "$EMU"                                     # the normal library, up to date, kept aside
cp "$EMU/libdm_emu.so" /tmp/libdm_emu_normal.so
trap 'cp /tmp/libdm_emu_normal.so "$EMU/libdm_emu.so"; make -s -C "$EMU"' EXIT
make -s -C "$EMU" OBJDIR=/tmp/emu_$KIND CXXFLAGS="-O1 -g $FLAGS -fno-omit-frame-pointer -std=c++17 -fPIC -DDM_EMU -I. -Wno-unknown-pragmas -w" -B libdm_emu.so
cd "$ROOT"
DM_ALLOW_EMULATOR=1 LD_PRELOAD=$RT python -m pytest tests -q -m "not gpu" -n 6 --deselect tests/test_native_caller.py "$@"
