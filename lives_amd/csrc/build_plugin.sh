#!/bin/bash
# Build livesgpu_fx.so (the weed plugin) against include/ only; it links liblivesgpu.so next to it.
set -e
cd "$(dirname "$0")"
gcc -O2 -std=c11 -Wall -Wextra -fPIC -shared -o ../livesgpu_fx.so fx_plugin.c -L.. -llivesgpu -Wl,-rpath,'$ORIGIN' -Wl,--no-undefined
echo "built $(cd .. && pwd)/livesgpu_fx.so"
