#!/bin/bash
# Build livesgpu_fx.so (the weed plugin) against include/ only; it links liblivesgpu.so next to it.
set -e
cd "$(dirname "$0")"
gcc -O2 -std=c11 -Wall -Wextra -fPIC -shared -o ../livesgpu_fx.so fx_plugin.c -L.. -llivesgpu -Wl,-rpath,'$ORIGIN' -Wl,--no-undefined
echo "built $(cd .. && pwd)/livesgpu_fx.so"
# liblivesgpu_dropin.so: the layer-op seam under the reference's own names (dropin.c), forwarding into liblivesgpu.so
gcc -O2 -std=c11 -Wall -Wextra -fPIC -shared -fvisibility=hidden -o ../liblivesgpu_dropin.so dropin.c -L.. -llivesgpu -Wl,-rpath,'$ORIGIN' -Wl,--no-undefined
echo "built $(cd .. && pwd)/liblivesgpu_dropin.so"
