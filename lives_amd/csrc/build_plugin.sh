#!/bin/bash
# Build livesgpu_fx.so (the weed plugin) against include/ only; it links liblivesgpu.so next to it.
set -e
cd "$(dirname "$0")"
gcc -O2 -std=c11 -Wall -Wextra -fPIC -shared -o ../livesgpu_fx.so fx_plugin.c -L.. -llivesgpu -Wl,-rpath,'$ORIGIN' -Wl,--no-undefined
echo "built $(cd .. && pwd)/livesgpu_fx.so"
# liblivesgpu_dropin.so: the layer-op seam under the reference's own names (dropin.c), forwarding into liblivesgpu.so
gcc -O2 -std=c11 -Wall -Wextra -fPIC -shared -fvisibility=hidden -o ../liblivesgpu_dropin.so dropin.c -L.. -llivesgpu -Wl,-rpath,'$ORIGIN' -Wl,--no-undefined
echo "built $(cd .. && pwd)/liblivesgpu_dropin.so"
# tools/libseam_host.so: the C render host of bench.py's seam_chain leg and tests/test_seam_host.py (tools/seam_host.c + tools/miniweed.c): links the REFERENCE
# names out of liblivesgpu_dropin.so, loads livesgpu_fx.so through weed_setup at run time
gcc -O2 -std=gnu11 -Wall -Wextra -fPIC -shared -o ../../tools/libseam_host.so ../../tools/seam_host.c ../../tools/miniweed.c -L.. -llivesgpu_dropin -llivesgpu \
  -Wl,-rpath,'$ORIGIN/../lives_amd' -Wl,--no-undefined -ldl -lpthread
echo "built $(cd ../../tools && pwd)/libseam_host.so"
